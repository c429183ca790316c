# values-only evaluation: persistent "dynamic" waves per CU of eval_values_kernel (of the 12 its registers allow); make TUNING=1
mkdir -p gpurun_out/r05e
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
for w in 8 6 10 12 4; do echo "TWR_VALUES_DYN_WPC=$w"; TWR_VALUES_DYN_WPC=$w python scripts/planner_split.py 128 1024 2>&1 | grep "B=" | cut -c1-60; done
