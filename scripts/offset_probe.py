"""The slow / fast state of the evaluation as a function of WHERE INSIDE ONE ALLOCATION the Jacobian buffer starts: one 9-GB
allocation, the C3 batch evaluated at different offsets into it.  If the state follows the offset, it is a property of the address
pattern (the kernels' concurrent streams against the memory's channel mapping), not of the allocation."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

dev = torch.device("cuda", 0)
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
xh = np.tile(base, (B // 256, 1)).reshape(-1)
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, dev, 0.5)
x = torch.from_numpy(xh).to(dev)
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
nj = int(batch.jac_off[-1])
big = torch.empty(nj + (2 << 27), dtype=torch.float64, device=dev)   # + 2 GiB of room


def timed(jptr):
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jptr, ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jptr, ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


print("allocation at 0x%x" % big.data_ptr())
for rep in range(2):
    for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 128 << 20, 512 << 20, 1 << 30, (1 << 30) + (2 << 20), 2 << 30):
        print("pass %d  offset %12d B (%8.2f MiB): %.3f ms/step" % (rep, off, off / 2**20, timed(big.data_ptr() + off)), flush=True)
