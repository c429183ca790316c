export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT
for p in "a TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "b TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" "c TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum"; do
  set -- $p; name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 scripts/alloc_states.py > $OUT/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/pmc_$name.log; continue; }
  echo "== pass $name: $@"; python3 scripts/alloc_states_summary.py $OUT/pmc_$name
done
