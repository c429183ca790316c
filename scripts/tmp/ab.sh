#!/bin/bash
for r in 1 2; do
for v in base new; do
  cp scripts/tmp/lib_$v.so towr_amd/libtowr_amd.so
  python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k.split('::')[1][:8]:round(v*1e3,1) for k,v in d['roofline']['path']['kernel_ms'].items()})"
done; done
