mkdir -p gpurun_out/r05a
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
for cfg in "12 1" "8 1" "16 1" "24 1" "12 0"; do set -- $cfg; echo "WPC=$1 FUSED=$2"; TWR_VALUES_WPC=$1 TWR_VALUES_FUSED=$2 python scripts/planner_split.py 128 1024 2>&1 | grep "B="; done
