#!/bin/bash
# A/B of candidate libraries against the working-tree library on ONE box: parity subset on the first candidate, then
# sweep (1024 candidates, per-kernel events) and C3.  Usage: bash scripts/r04_ab3.sh <tag> cand.so [cand2.so ...]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
TWR_AMD_LIB=$PWD/towr_amd/$1 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "baseline_configs or sweep_c4 or ragged_batch or fused_launch or full_size or shared_by_content" --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
rc=$?
tail -3 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" $OUT/gpu_tests.log | head -40; exit 1; fi
echo "--- sweep 1024"
AB_REPS=${AB_REPS:-4} timeout -k 10 600 python3 scripts/ab.py libtowr_amd.so "$@" -- --workload sweep --batch 1024 2>&1 | tee $OUT/ab_sweep.txt
echo "--- C3 8192"
AB_REPS=3 timeout -k 10 500 python3 scripts/ab.py libtowr_amd.so "$@" 2>&1 | tee $OUT/ab_c3.txt
