#!/bin/bash
# per-kernel durations of one bench run (rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
OUT=gpurun_out/ktrace_${1:-run}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $OUT/bench.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'twr::' in r['Name']:
        print("%-60s calls %s avg %.1f us  min %.1f max %.1f"%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3))
PY
tail -1 $OUT/bench.log | cut -c1-300
