#!/bin/bash
# A/B of candidate libraries on the optimised-timings workload (2048 problems) and towr's whole constraint list: parity tests that
# touch the pre-pass / sampling kernels on the first candidate, then per-kernel times.  Usage: bash scripts/r04_ab_timings.sh <tag> cand.so ...
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
TWR_AMD_LIB=$PWD/towr_amd/$1 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "timings or sampling or initial_guess or contact_plan or scores or hostile or whole_default or persistent_node or random_structures" --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
rc=$?
tail -3 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" $OUT/gpu_tests.log | head -40; exit 1; fi
echo "--- timings 2048"
AB_REPS=${AB_REPS:-3} timeout -k 10 600 python3 scripts/ab.py libtowr_amd.so "$@" -- --sets timings --batch 2048 2>&1 | tee $OUT/ab_timings.txt
echo "--- all sets 8192"
AB_REPS=2 timeout -k 10 600 python3 scripts/ab.py libtowr_amd.so "$@" -- --sets all 2>&1 | tee $OUT/ab_all.txt
