#!/bin/bash
# Round-3 opening measurements on one box: the counter listing of gfx950 (profiler-hang diagnosis), the sweep workload's
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), and C3 at the sweep's batch size for comparison.
export TMPDIR=/tmp
OUT=gpurun_out/r03_base
mkdir -p $OUT
timeout -k 10 120 rocprofv3 -L > $OUT/rocprof_L.txt 2>&1
echo "listing done" 
B="--steps 30 --warmup 5 --no-cpu-baseline --no-scale-c5 --no-timings-c3"
python3 bench.py $B --workload sweep --batch 1024 > $OUT/sweep1024.json 2> $OUT/sweep1024.err || exit 1
python3 bench.py $B --batch 1024 > $OUT/c3_1024.json 2> $OUT/c3_1024.err || exit 1
python3 bench.py $B --sets timings --batch 2048 > $OUT/timings2048.json 2> $OUT/timings2048.err || exit 1
echo "bench done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/sweep_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --workload sweep --batch 1024 > $OUT/sweep_$c.log 2>&1 || exit 1
  echo "pmc $c done"
done
python3 - <<'PY'
import csv, glob, collections, json
out = "gpurun_out/r03_base"
acc = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/sweep_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "twr::" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("twr::")[1].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
res = {}
for (k, n), v in acc.items():
    res.setdefault(k, {})[n] = sum(v) / len(v)
for k, d in res.items():
    d["hbm_bytes"] = (d.get("WRITE_SIZE", 0) + 2 * d.get("FETCH_SIZE", 0)) * 1024
json.dump(res, open(out + "/sweep_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
