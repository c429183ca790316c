#!/bin/bash
# Round-3 profile set, all taken at one tree state: rocprofv3 kernel trace of the default bench (hot sets) and of
# --sets timings, PMC passes (SQ groups, FETCH_SIZE, WRITE_SIZE: one counter group per pass, with --kernel-trace only),
# HBM traffic of the timings and the sweep workloads, and profiles/traffic.json stamped with the hash of the kernel
# sources.  Usage (on the GPU box): bash scripts/profile_r03.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="--steps 20 --warmup 3 --no-cpu-baseline --no-scale-c5 --no-timings-c3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 bench.py $B > $OUT/bench_under_rocprof.json 2> $OUT/ktrace.err || exit 1
echo "ktrace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace_t -- python3 bench.py $B --sets timings --batch 2048 > $OUT/bench_timings_under_rocprof.json 2> $OUT/ktrace_t.err || exit 1
echo "ktrace timings done"
pmc() { # name bench-args counters...
  name=$1; shift; extra=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 $extra > $OUT/$name.log 2>&1 || return 1
  echo "pmc $name done"
}
pmc sq1 "" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
pmc sq2 "" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM || exit 1
pmc tcc1 "" FETCH_SIZE || exit 1
pmc tcc2 "" WRITE_SIZE || exit 1
pmc sq1_t "--sets timings --batch 2048" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
pmc sq2_t "--sets timings --batch 2048" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM || exit 1
pmc tcc1_t "--sets timings --batch 2048" FETCH_SIZE || exit 1
pmc tcc2_t "--sets timings --batch 2048" WRITE_SIZE || exit 1
pmc tcc1_s "--workload sweep --batch 1024" FETCH_SIZE || exit 1
pmc tcc2_s "--workload sweep --batch 1024" WRITE_SIZE || exit 1
python3 scripts/profile_summary.py $OUT $TAG
