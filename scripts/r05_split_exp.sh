# the fused launch's residency split (TWR_FUSED_SPLIT = eighths for the rom role; 8 = unsplit) at shard sizes (make TUNING=1)
mkdir -p gpurun_out/r05a
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
for sp in 0 5 4 6; do echo "SPLIT=$sp (0 = the library's rule)"; TWR_FUSED_SPLIT=$sp python scripts/small_batches.py 192 256 320 384 512 768 1024 2>&1 | grep "us/step"; done
