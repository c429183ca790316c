#!/bin/bash
# usage: sweep_mix.sh "dyn:rom dyn:rom ..."   (blocks per CU of each kernel; s<d>:<r> = serial)
export TMPDIR=/tmp
for cfg in $1; do
  d=${cfg%%:*}; r=${cfg##*:}
  if [[ $d == s* ]]; then export TWR_SERIAL=1; d=${d#s}; else unset TWR_SERIAL; fi
  export TWR_DYN_BPC=$d TWR_ROM_BPC=$r
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg $cfg: %.1f us/step  %.3e cb/s  frac %.3f'%(d['ms_per_step']*1e3,d['value'],d['roofline']['frac']))"
done
