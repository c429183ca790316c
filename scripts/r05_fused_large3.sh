mkdir -p gpurun_out/r05g
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
for cfg in "640 384" "704 320" "608 416" "640 768" "672 704" "640 640" "640 1024" "512 1024" "768 512" "576 448"; do set -- $cfg; echo -n "GROM=$1 GDYN=$2: "; TWR_FUSED_MAX_ROM=400000 TWR_FUSED_GROM=$1 TWR_FUSED_GDYN=$2 python scripts/c3_fused_vs_separate.py 2>&1 | grep "three launches"; done
