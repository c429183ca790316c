#!/bin/bash
# SQ counters of the values-only kernels (scripts/values_c3.py under rocprofv3 --pmc, two passes)
export TMPDIR=/tmp
OUT=gpurun_out/values_pmc
mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES \
  --output-format csv -d $OUT/a -- python3 scripts/values_c3.py 1 > $OUT/a.log 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT/b -- python3 scripts/values_c3.py 1 > $OUT/b.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("gpurun_out/values_pmc/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "values" not in k and "node_kernel2" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
