#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03_abl
mkdir -p $OUT
export TWR_PDYN_LDS_KB=32 TWR_PDYN_NODES=15 TWR_PDYN_BPC=4
for f in 0 256 512 768 1024 2048 3840 1792; do
  TWR_DEBUG_FLAGS=$f timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scale-c5 --sets timings --batch 2048 > $OUT/b_$f.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  python3 -c "
import json
d=json.load(open('$OUT/b_$f.json'))
print('flags $f', {k.split('::')[1]: round(v,4) for k,v in d['roofline']['path']['kernel_ms'].items()})"
done
