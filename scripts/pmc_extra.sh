#!/bin/bash
# Extra SQ counters of the default bench (one --pmc pass, never combined with other trace domains): waits on LDS, the
# accumulated number of vector-memory / LDS / scalar-memory instructions in flight (LEVEL / instructions = average
# latency in the counter's cycle unit).  NOTE: the round-2 pass that also asked for a list of TA_* / TCP_* stall counters
# exceeded what one pass can collect (rocprofiler error 38 -> abort inside the first HIP call -> rocprofv3's abort
# handler never exits: a silent process until the watchdog).  Other blocks go into passes of their own, few counters
# each, under `timeout -k 10 240`.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_extra
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 > $OUT/a.log 2>&1 || exit 1
echo done
