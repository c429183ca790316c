#!/bin/bash
# full GPU test tier + the default bench (summary of the legs on stdout)
export TMPDIR=/tmp
bash scripts/gpu_tests.sh 900 > /dev/null; tail -2 gpurun_out/gpu_tests.log
python3 bench.py --steps ${1:-30} --warmup 5 ${2:---no-cpu-baseline} > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err || tail -5 gpurun_out/bench_check.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_check.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"]), {k.split("::")[1]: round(v, 4) for k, v in r["path"]["kernel_ms"].items()}, "dominant", round(r["frac"], 3), "path", round(r["path"]["frac"], 3), "traffic", r["traffic"])
for leg in ("timings_c3", "all_sets_c3"):
    t = d.get(leg)
    if t:
        print(leg, round(t["value"]), {k: round(v, 4) for k, v in t["roofline"]["path"]["kernel_ms"].items()}, "path", round(t["roofline"]["path"]["frac"], 3), "traffic", t["roofline"]["traffic"])
c = d.get("scale_c5")
if c:
    print("c5", round(c["value"]), "traffic_ratio", c["traffic_ratio"], "planner", round(c["planner"]["value"]))
if "cpu_baseline" in d:
    print("cpu", round(d["cpu_baseline"]["value"], 1), d["cpu_baseline"]["all_cores"])
PY
