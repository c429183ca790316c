# the fused launch (roles co-resident) at LARGE batches of one structure: is the three-launch path still ahead at 4096 / 8192?
mkdir -p gpurun_out/r05b
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
echo "three launches (library rule)"; python scripts/c3_chunks.py 2048 4096 8192
for cfg in "512 1024" "576 896" "640 768" "704 640" "448 1152"; do set -- $cfg; echo "fused, GROM=$1 GDYN=$2 (blocks; a dyn block = 2 waves)"; TWR_FUSED_MAX_ROM=400000 TWR_FUSED_GROM=$1 TWR_FUSED_GDYN=$2 python scripts/c3_chunks.py 2048 8192; done
