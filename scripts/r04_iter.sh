#!/bin/bash
# One iteration of the round-4 kernel work on one box: GPU parity tier, then the workloads being moved
# (1024-candidate sweep with per-kernel events, small ragged batches, the default C3 workload without its extra legs).
# Usage: bash scripts/r04_iter.sh <tag> [pytest -k expression]
export TMPDIR=/tmp
TAG=${1:-iter}
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
if [ -n "${2:-}" ]; then K=(-k "$2"); else K=(); fi
timeout -k 10 600 python -m pytest tests -x -q -m gpu "${K[@]}" --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
rc=$?
tail -3 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" $OUT/gpu_tests.log | head -40; exit 1; fi
B="--steps 30 --warmup 5 --no-cpu-baseline --no-scale-c5 --no-timings-c3"
timeout -k 10 200 python3 bench.py $B --workload sweep --batch 1024 > $OUT/sweep1024.json 2> $OUT/sweep1024.err || exit 1
timeout -k 10 200 python3 scripts/small_batches.py 128 256 512 1024 > $OUT/small_batches.txt 2>&1 || exit 1
cat $OUT/small_batches.txt
timeout -k 10 300 python3 bench.py $B > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
python3 - $OUT <<'PY'
import json, sys
for f in ("sweep1024", "bench_c3"):
    d = json.loads(open("%s/%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    print(f, "%.3f M cb/s" % (d["value"] / 1e6), {k: round(v, 4) for k, v in d["roofline"]["path"]["kernel_ms"].items()}, "path %.3f" % d["roofline"]["path"]["frac"])
PY
