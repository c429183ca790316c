#!/bin/bash
# PMC passes (SQ groups) for the optimised-timings kernels
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${1:-timings}
mkdir -p $OUT
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --sets timings --batch 2048 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    kn=r['Kernel_Name']
    if 'twr::' in kn:
        short=kn.split('twr::')[1].split('(')[0]
        acc[(short,r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print("%-22s %-26s mean/launch %.6g"%(k[0],k[1],sum(v)/len(v)))
PY
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
