mkdir -p gpurun_out/r05b
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
echo "three launches (library rule)"; python scripts/c3_chunks.py 8192
for cfg in "640 768" "640 896" "640 640" "608 832" "672 704" "624 800" "656 736" "640 1024" "576 1152"; do set -- $cfg; echo "fused, GROM=$1 GDYN=$2"; TWR_FUSED_MAX_ROM=400000 TWR_FUSED_GROM=$1 TWR_FUSED_GDYN=$2 python scripts/c3_chunks.py 8192 2>&1 | grep "B="; done
echo "three launches again"; python scripts/c3_chunks.py 8192 2>&1 | grep "B="
