#!/bin/bash
# convenience: run the GPU test tier on the gpurun box (progress goes to gpurun_out/ as it happens)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 ${1:-600} python -m pytest tests -x -v -m gpu --timeout 900 --durations=6 -p no:cacheprovider 2>&1 | tee gpurun_out/gpu_tests.log | grep -E "PASS|FAIL|ERROR|passed|failed|Timeout|error|s call" 
