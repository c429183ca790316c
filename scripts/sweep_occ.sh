#!/bin/bash
# usage: sweep_occ.sh "4 6 8"
export TMPDIR=/tmp
for k in $1; do
  export TWR_BLOCKS_PER_CU=$k
  OUT=gpurun_out/occ_$k; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "blocks/CU=$k: $(python3 -c "
import csv,sys
print(' '.join('%s=%.0fus(min %.0f)'%(r['Name'].split('twr::')[1][:10],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3) for r in csv.DictReader(open('$f')) if 'twr::' in r['Name']))")"
done
