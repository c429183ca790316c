# (historical: the TWR_FLAT_DIAG hooks lived in the persistent form of the values-only kernels, which was not kept)
# values-only kernels, what the time is made of: diagnostic builds (WRONG RESULTS on purpose) 1: no math, 2: no sin / cos, 3: x loaded once per wave
mkdir -p gpurun_out/r05e
for d in 0 1 2 3; do
  make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc DIAG=-DTWR_FLAT_DIAG=$d > /dev/null 2>&1 || exit 1
  echo "TWR_FLAT_DIAG=$d"; python scripts/values_c3.py 1 2>&1 | grep -E "values_c3|Error|error" | cut -c1-220
done
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc > /dev/null 2>&1
