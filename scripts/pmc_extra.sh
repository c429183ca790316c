#!/bin/bash
# Extra SQ counters of the default bench (one --pmc pass, never combined with other trace domains): waits on LDS, the
# accumulated number of vector-memory / LDS / scalar-memory instructions in flight (LEVEL / instructions = average
# latency in the counter's cycle unit).  NOTE: a pass with TA_* / TCP_* stall counters hung the run on this pool
# (killed after 7 silent minutes, no strike) -- they are deliberately not collected here.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_extra
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 > $OUT/a.log 2>&1 || exit 1
echo done
