#!/bin/bash
# A/B of two library builds over sweep shard sizes on one box.  Usage: bash scripts/r04_even.sh <tag> libA.so libB.so
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
for B in 256 384 512 768 1024; do
  echo "--- sweep $B"
  timeout -k 10 300 python3 scripts/ab.py "$@" -- --workload sweep --batch $B 2>&1 | tee -a $OUT/ab_sweep.txt || exit 1
done
echo "--- C3"
timeout -k 10 300 python3 scripts/ab.py "$@" 2>&1 | tee -a $OUT/ab_c3.txt
