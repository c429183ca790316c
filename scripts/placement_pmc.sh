#!/bin/bash
# address-translation counters of the fused sweep launch per allocation of its buffers (scripts/placement_probe5.py under
# rocprofv3 --pmc): does a slow allocation miss more in the translation caches?
export TMPDIR=/tmp
OUT=gpurun_out/placement_pmc
rm -rf $OUT; mkdir -p $OUT
L="0 2 2 0 2 4 1 2 6 2 1 1 0 0"
PROBE_STEPS=10 timeout -k 10 280 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/a -- python3 scripts/placement_probe5.py $L > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
grep ballast $OUT/a.log | cut -c1-60
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/placement_pmc/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_fused_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob("gpurun_out/placement_pmc/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_fused_kernel" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(rows)
per = 30   # 20 warm-up + 10 timed dispatches per allocation
for a in range(len(ids) // per):
    chunk = ids[a * per + 20:(a + 1) * per]
    avg = lambda k: sum(rows[i].get(k, 0.0) for i in chunk) / len(chunk)
    print("allocation %2d: kernel %.1f us  UTCL1 miss %.3g hit %.3g req %.3g  UTCL2 busy %.3g / active %.3g" % (
        a, sum(dur.get(i, 0.0) for i in chunk) / len(chunk), avg("TCP_UTCL1_TRANSLATION_MISS_sum"), avg("TCP_UTCL1_TRANSLATION_HIT_sum"),
        avg("TCP_UTCL1_REQUEST_sum"), avg("GRBM_UTCL2_BUSY"), avg("GRBM_GUI_ACTIVE")))
PY
