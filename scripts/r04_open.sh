#!/bin/bash
# Round-4 opening call on one box: the GPU test tier, then the sweep baselines this round wants to move
# (1024-candidate sweep with per-kernel events, small ragged batches), then the default bench line.
export TMPDIR=/tmp
OUT=gpurun_out/r04_open
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests -x -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1 || { tail -30 $OUT/gpu_tests.log; exit 1; }
tail -3 $OUT/gpu_tests.log
B="--steps 30 --warmup 5 --no-cpu-baseline --no-scale-c5 --no-timings-c3"
timeout -k 10 200 python3 bench.py $B --workload sweep --batch 1024 > $OUT/sweep1024.json 2> $OUT/sweep1024.err || exit 1
timeout -k 10 200 python3 scripts/small_batches.py 128 256 512 1024 > $OUT/small_batches.txt 2>&1 || exit 1
cat $OUT/small_batches.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python3 - <<'PY'
import json
for f in ("sweep1024", "bench_default"):
    d = json.loads(open("gpurun_out/r04_open/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "%.3f M cb/s" % (d["value"] / 1e6), d["roofline"]["path"]["kernel_ms"], "path %.3f" % d["roofline"]["path"]["frac"])
    if "scale_c5" in d:
        print(" scale_c5 %.3f M  timings %.3f M  all_sets %.3f M" % (d["scale_c5"]["value"] / 1e6, d["timings_c3"]["value"] / 1e6, d["all_sets_c3"]["value"] / 1e6))
PY
