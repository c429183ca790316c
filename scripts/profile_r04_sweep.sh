#!/bin/bash
# The sweep passes of scripts/profile_r04.sh alone (kernel trace, FETCH_SIZE, WRITE_SIZE, SQ group 1), into the same directory:
# re-taken after a HOST-side change that alters what the sweep launches (the kernels' source hash is unchanged).
export TMPDIR=/tmp
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="--steps 20 --warmup 3 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --workload sweep --batch 1024"
rm -rf $OUT/ktrace_s $OUT/tcc1_s $OUT/tcc2_s $OUT/sq1_s
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace_s -- python3 bench.py $B > $OUT/bench_ktrace_s.json 2> $OUT/ktrace_s.err || exit 1
for p in "tcc1_s FETCH_SIZE" "tcc2_s WRITE_SIZE" "sq1_s SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  set -- $p; name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --workload sweep --batch 1024 > $OUT/$name.log 2>&1 || exit 1
  echo "pmc $name done"
done
