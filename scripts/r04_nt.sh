#!/bin/bash
# Non-temporal copy-out stores, per kernel family: lib_ntd.so (dyn_kernel only), lib_ntdnp.so (+ node_chunk_kernel + the
# optimised-timings kernels), lib_ntall.so (+ rom_kernel) against the working-tree library on ONE box: sweep 1024 (events),
# small batches (no events: the fused launch), C3, --sets timings, --sets all.
export TMPDIR=/tmp
OUT=gpurun_out/r04_nt
mkdir -p $OUT
LIBS="libtowr_amd.so lib_ntd.so lib_ntdnp.so lib_ntall.so"
TWR_AMD_LIB=$PWD/towr_amd/lib_ntall.so timeout -k 10 600 python -m pytest tests -x -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
rc=$?
tail -2 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" $OUT/gpu_tests.log | head -40; exit 1; fi
echo "--- sweep 1024"; AB_REPS=2 timeout -k 10 600 python3 scripts/ab.py $LIBS -- --workload sweep --batch 1024 2>&1 | tee $OUT/ab_sweep.txt
echo "--- C3 8192"; AB_REPS=2 timeout -k 10 600 python3 scripts/ab.py $LIBS 2>&1 | tee $OUT/ab_c3.txt
echo "--- timings 2048"; AB_REPS=2 timeout -k 10 600 python3 scripts/ab.py libtowr_amd.so lib_ntdnp.so -- --sets timings --batch 2048 2>&1 | tee $OUT/ab_timings.txt
echo "--- all sets 8192"; AB_REPS=2 timeout -k 10 600 python3 scripts/ab.py libtowr_amd.so lib_ntdnp.so lib_ntd.so -- --sets all 2>&1 | tee $OUT/ab_all.txt
echo "--- small batches (no events)"
for lib in libtowr_amd.so lib_ntd.so lib_ntall.so libtowr_amd.so lib_ntd.so lib_ntall.so; do
  echo $lib; TWR_AMD_LIB=$PWD/towr_amd/$lib timeout -k 10 200 python3 scripts/small_batches.py 64 128 256 512 1024 2>&1 | grep "us/step" | tee -a $OUT/small_$lib.txt
done
