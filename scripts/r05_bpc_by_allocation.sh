# persistent workgroups per CU of rom_kernel / dyn_kernel on several allocations of the C3 buffers in one process; make TUNING=1
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
python scripts/c3_bpc_by_allocation.py 2>&1 | grep ballast
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc > /dev/null 2>&1
