# node_chunk_kernel: persistent waves per CU (TWR_NODE_BPC, default 16) on the default towr list, 8192 C3 problems; make TUNING=1
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc TUNING=1 > /dev/null 2>&1
for r in 1 2; do for w in 16 12 20 24 32; do echo "TWR_NODE_BPC=$w"; TWR_NODE_BPC=$w python bench.py --sets all --steps 20 --warmup 3 --no-cpu-baseline --no-scale-c5 --no-timings-c3 --no-values-c3 --placement-tries 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k.split('::')[1]: round(v,4) for k,v in d['roofline']['path']['kernel_ms'].items()})"; done; done
make -C towr_amd/csrc clean > /dev/null; make -C towr_amd/csrc > /dev/null 2>&1
