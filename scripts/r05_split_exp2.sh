# the new split rule on batches that SHARE one structure (C3, plain stores), and the fused launch beyond its threshold
mkdir -p gpurun_out/r05a
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
echo "C3 one structure: unsplit (TWR_FUSED_SPLIT=8)"; TWR_FUSED_SPLIT=8 python scripts/c3_chunks.py 128 256 320 512 640 1024 2048
echo "C3 one structure: library rule"; python scripts/c3_chunks.py 128 256 320 512 640 1024 2048
echo "C3 one structure: library rule, fused up to 40000 rom slices"; TWR_FUSED_MAX_ROM=40000 python scripts/c3_chunks.py 640 1024 2048
echo "sweep: library rule"; python scripts/small_batches.py 128 256 512 1024 2>&1 | grep "us/step"
echo "sweep: three launches (TWR_FUSED_MAX_ROM=1)"; TWR_FUSED_MAX_ROM=1 python scripts/small_batches.py 512 1024 2>&1 | grep "us/step"
