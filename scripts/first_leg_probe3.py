"""Does WHERE the output buffers sit in device memory change the store rate?  Same C3 batch; before the x / g / jac buffers are
allocated a ballast of G GB is allocated (and kept), which pushes them to other physical addresses."""
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


def region(what):
    x = torch.from_numpy(xh).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    out = []
    for rep in range(2):
        batch.profile_begin(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        k, _ = batch.profile_end()
        out.append("%.3f (dyn %.3f rom %.3f)" % (ms, k["dynamic"], k["rangeofmotion"]))
    print("%-28s %s   jac at 0x%x" % (what, "  ".join(out), j.data_ptr()), flush=True)
    del x, g, j
    torch.cuda.empty_cache()


device_power_warmup(torch, dev, 0.5)
region("no ballast")
for gb in (1, 2, 4, 7, 14, 28, 56, 112):
    ballast = torch.empty(gb * (1 << 27), dtype=torch.float64, device=dev)
    region("ballast %3d GB" % gb)
    del ballast
    torch.cuda.empty_cache()
region("no ballast again")
