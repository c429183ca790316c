# explicit grids of the fused launch's two roles at small shard sizes (make TUNING=1 build; TWR_FUSED_GROM / TWR_FUSED_GDYN)
mkdir -p gpurun_out/r05a
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
echo "default"; python scripts/small_batches.py 128 256 2>&1 | grep "us/step"
for cfg in "640 384" "683 341" "688 336" "704 320" "768 256" "600 424" "512 512" "1024 344" "1024 512" "683 512" "820 408"; do set -- $cfg; echo "GROM=$1 GDYN=$2"; TWR_FUSED_GROM=$1 TWR_FUSED_GDYN=$2 python scripts/small_batches.py 128 256 2>&1 | grep "us/step"; done
