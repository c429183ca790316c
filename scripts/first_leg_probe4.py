"""Placement of the output buffers, follow-up: the ballast is FREED again before the steps are timed (the buffers stay where they
were put); several ballast sizes, twice, to see whether the pattern is stable within a process."""
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


def region(what, gb):
    ballast = torch.empty(max(1, int(gb * (1 << 27))), dtype=torch.float64, device=dev)
    x = torch.from_numpy(xh).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    del ballast
    torch.cuda.empty_cache()
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("%-34s %.3f ms/step   jac at 0x%x" % (what, ms, j.data_ptr()), flush=True)
    del x, g, j
    torch.cuda.empty_cache()


device_power_warmup(torch, dev, 0.5)
for rep in range(2):
    for gb in (0, 0.5, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20):
        region("pass %d, ballast %4.1f GB (freed)" % (rep, gb), gb)
