# what can removing the LDS bank conflicts of the image scatter buy?  A/B on ONE box, alternating: the product
# library against the diagnostic build -DTWR_DIAG_NOCONFLICT (every scattered ds_write_b64 of dyn_kernel / rom_kernel goes to a
# conflict-free address, no address arithmetic; results wrong on purpose), C3 8192 problems, per-kernel events.  Then one PMC
# pass on each build (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) to show that the conflicts are really gone.
export TMPDIR=/tmp
OUT=gpurun_out/r05_lds
mkdir -p $OUT
B="--steps 20 --warmup 5 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --no-values-c3"
mkdir -p /tmp/libs
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc DIAG=-DTWR_DIAG_NOCONFLICT > /dev/null 2>&1 || exit 1
cp towr_amd/libtowr_amd.so /tmp/libs/diag.so
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc > /dev/null 2>&1 || exit 1
cp towr_amd/libtowr_amd.so /tmp/libs/prod.so
for rep in 1 2 3; do
  for v in prod diag; do
    cp /tmp/libs/$v.so towr_amd/libtowr_amd.so
    python3 bench.py $B > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err || exit 1
    python3 -c "
import json,sys
d=json.load(open('$OUT/bench_${v}_$rep.json'))
print('$v $rep', round(d['value']/1e6,3), 'M cb/s', {k:round(x,4) for k,x in d['roofline']['path']['kernel_ms'].items()})"
  done
done
for v in prod diag; do
  cp /tmp/libs/$v.so towr_amd/libtowr_amd.so
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --no-values-c3 > $OUT/pmc_$v.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
for k, c in acc.items():
    if "dyn_kernel" in k or "rom_kernel" in k:
        print("$v", k[:60], "launches", n[k], "conflict/active %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)), "LDS insts/launch %.3e VALU/launch %.3e" % (c["SQ_INSTS_LDS"] / max(n[k], 1), c["SQ_INSTS_VALU"] / max(n[k], 1)))
PY
done
cp /tmp/libs/prod.so towr_amd/libtowr_amd.so
