#!/bin/bash
# A/B on ONE box with the parity tier run on the FIRST candidate library (TWR_AMD_LIB), then sweep + C3 kernel times of
# all libraries (alternating).  Usage: bash scripts/r04_ab2.sh <tag> "<pytest -k or empty>" base.so cand.so ...
export TMPDIR=/tmp
TAG=$1; KEXPR=$2; shift 2
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
if [ -n "$KEXPR" ]; then K=(-k "$KEXPR"); else K=(); fi
TWR_AMD_LIB=$PWD/towr_amd/$2 timeout -k 10 600 python -m pytest tests -x -q -m gpu "${K[@]}" --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
rc=$?
tail -3 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" $OUT/gpu_tests.log | head -40; exit 1; fi
echo "--- C3 8192"
timeout -k 10 500 python3 scripts/ab.py "$@" 2>&1 | tee $OUT/ab_c3.txt
echo "--- sweep 1024"
timeout -k 10 400 python3 scripts/ab.py "$@" -- --workload sweep --batch 1024 2>&1 | tee $OUT/ab_sweep.txt
