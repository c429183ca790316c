# (historical: the persistent form of the values-only kernels, not kept -- its knobs TWR_VALUES_WPC / TWR_VALUES_DYN_WEIGHT are no longer read)
# values-only evaluation, persistent flat kernels: resident waves per CU and the dyn : rom share of the one-launch form; make TUNING=1
mkdir -p gpurun_out/r05e
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
for w in 12 11 10 8 6; do echo "TWR_VALUES_WPC=$w"; TWR_VALUES_WPC=$w python scripts/values_c3.py 1 2>&1 | grep values_c3; done
for dw in 5 6 7 8; do echo "TWR_VALUES_DYN_WEIGHT=$dw : 3"; TWR_VALUES_DYN_WEIGHT=$dw python scripts/values_c3.py 1 2>&1 | grep values_c3; done
for w in 12 10 8; do echo "TWR_VALUES_WPC=$w"; TWR_VALUES_WPC=$w python scripts/planner_split.py 128 1024 2>&1 | grep "B=" | cut -c1-60; done
