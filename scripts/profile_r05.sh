#!/bin/bash
# Round-5 profile set, all taken at one tree state on one box: rocprofv3 kernel traces (--stats) of the default bench (C3
# hot sets), of --sets timings and of the 1024-candidate sweep; PMC passes (one counter group per pass, --pmc only ever
# with --kernel-trace, every pass under `timeout -k 10 240`: an over-full counter request makes rocprofv3 hang in its abort
# handler) for SQ groups + FETCH_SIZE + WRITE_SIZE of C3, and FETCH_SIZE / WRITE_SIZE of --sets timings, --sets all and
# the sweep.  scripts/profile_summary.py condenses the merged output into profiles/ (run it here, after gpurun).
# Usage (on the GPU box): bash scripts/profile_r05.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="--steps 20 --warmup 3 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --no-values-c3"
kt() { # name bench-args
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py $B "$@" > $OUT/bench_$name.json 2> $OUT/$name.err || return 1
  echo "ktrace $name done"
}
kt ktrace || exit 1
kt ktrace_t --sets timings --batch 2048 || exit 1
kt ktrace_s --workload sweep --batch 1024 || exit 1
kt ktrace_a --sets all || exit 1
pmc() { # name bench-args counters...
  name=$1; shift; extra=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --no-values-c3 $extra > $OUT/$name.log 2>&1 || return 1
  echo "pmc $name done"
}
pmc sq1 "" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
pmc sq2 "" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM || exit 1
pmc tcc1 "" FETCH_SIZE || exit 1
pmc tcc2 "" WRITE_SIZE || exit 1
pmc tcc1_t "--sets timings --batch 2048" FETCH_SIZE || exit 1
pmc tcc2_t "--sets timings --batch 2048" WRITE_SIZE || exit 1
pmc tcc1_s "--workload sweep --batch 1024" FETCH_SIZE || exit 1
pmc tcc2_s "--workload sweep --batch 1024" WRITE_SIZE || exit 1
pmc tcc1_a "--sets all" FETCH_SIZE || exit 1
pmc tcc2_a "--sets all" WRITE_SIZE || exit 1
pmc sq1_s "--workload sweep --batch 1024" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
# round 5: the values-only leg (eval_values_kernel; with per-kernel events values_flat_kernel): kernel trace + VALU instruction counts
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace_v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 > $OUT/bench_ktrace_v.json 2> $OUT/ktrace_v.err || exit 1
echo "ktrace ktrace_v done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq_v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 > $OUT/sq_v.log 2>&1 || exit 1
echo "pmc sq_v done"
echo "profile set $TAG complete"
