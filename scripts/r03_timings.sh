#!/bin/bash
# optimised-timings path: parity tests, then the --sets timings bench (2048 problems) for a list of LDS budgets
export TMPDIR=/tmp
OUT=gpurun_out/r03_tim
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "timings or optimised or edge or fuzz or ragged or mixed" -p no:cacheprovider > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-scale-c5 --sets timings --batch 2048 > $OUT/bench_$(echo $cfg | tr ' =' '__').json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 - "$cfg" $OUT/bench_$(echo $cfg | tr ' =' '__').json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], round(d["value"]), "cb/s", {k.split("::")[1]: round(v, 4) for k, v in d["roofline"]["path"]["kernel_ms"].items()}, "path", round(d["roofline"]["path"]["frac"], 3))
PY
done
