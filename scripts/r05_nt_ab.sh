make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
python scripts/c3_nt_ab.py 2>&1 | grep -v amdgpu
