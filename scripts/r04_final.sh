#!/bin/bash
# GPU tier on the working-tree library, then the working-tree library against lib_base.so (the tree at the start of this
# session): sweep 1024 with per-kernel events, small batches without, C3.  Usage: bash scripts/r04_final.sh <tag>
export TMPDIR=/tmp
TAG=${1:-final}
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
rc=$?
tail -3 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" $OUT/gpu_tests.log | head -40; exit 1; fi
echo "--- sweep 1024 (per-kernel events)"
AB_REPS=3 timeout -k 10 400 python3 scripts/ab.py lib_base.so libtowr_amd.so -- --workload sweep --batch 1024 2>&1 | tee $OUT/ab_sweep.txt
echo "--- small batches (no events)"
for rep in 1 2; do
  for lib in lib_base.so libtowr_amd.so; do
    echo $lib; TWR_AMD_LIB=$PWD/towr_amd/$lib timeout -k 10 200 python3 scripts/small_batches.py 128 256 384 512 768 1024 2>&1 | grep "us/step" | tee -a $OUT/small_$lib.txt
  done
done
echo "--- C3 8192"
AB_REPS=3 timeout -k 10 500 python3 scripts/ab.py lib_base.so libtowr_amd.so 2>&1 | tee $OUT/ab_c3.txt
