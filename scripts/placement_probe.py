"""Is the placement effect (first_leg_probe*.py) a property of the MEMORY or of our kernels?  For several placements of the same
6.7-GB output buffer: the time of a plain torch.fill_ of it (one dense store stream, nothing of ours) beside the time of a C3
evaluation step into it."""
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
for gb in (0, 2, 5, 10, 0, 20, 0):
    ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb else None
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    del ballast
    for _ in range(3):
        j.fill_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        j.fill_(1.0)
    torch.cuda.synchronize()
    fill_ms = (time.perf_counter() - t0) / 10 * 1e3
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("jac behind %4.1f GB of ballast at 0x%x: fill_ %.3f ms = %.2f TB/s    evaluation step %.3f ms" % (gb, j.data_ptr(), fill_ms, j.numel() * 8 / fill_ms / 1e9, ms), flush=True)
    del j
    torch.cuda.empty_cache()
