"""C3, 8192 problems, ONE process, SEVERAL allocations of the buffers (behind ballasts of different sizes): plain copy-out stores
against non-temporal ones on each (TUNING build: TWR_STREAM_NT is read when a batch is created), per-kernel events -- does the
store policy matter differently on a slow and on a fast allocation?"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

dev = torch.device("cuda", 0)
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
os.environ["TWR_STREAM_NT"] = "0"
plain = ta.Batch([S], [0] * B, device=0)
os.environ["TWR_STREAM_NT"] = "1"
nt = ta.Batch([S], [0] * B, device=0)
assert not plain.streaming_stores() and nt.streaming_stores()
base = perturbed_inputs(S, model, 256, 0)
xh = np.tile(base, (B // 256, 1)).reshape(-1)
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, dev, 0.5)


def run(batch, x, g, j):
    batch.profile_begin(20)
    torch.cuda.synchronize()
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    k, _ = batch.profile_end()
    return k


for gb in [float(a) for a in sys.argv[1:]] or [0, 2, 5, 10, 1, 3, 7, 14, 0, 2]:
    ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb > 0 else None
    x = torch.from_numpy(xh).to(dev)
    g = torch.empty(int(plain.g_off[-1]), dtype=torch.float64, device=dev)
    j = torch.empty(int(plain.jac_off[-1]), dtype=torch.float64, device=dev)
    del ballast
    run(plain, x, g, j); run(nt, x, g, j)
    a, b, c, d = run(plain, x, g, j), run(nt, x, g, j), run(plain, x, g, j), run(nt, x, g, j)
    print("ballast %4.1f GB: plain dyn %.3f / %.3f rom %.3f / %.3f    nt dyn %.3f / %.3f rom %.3f / %.3f" % (
        gb, a["dynamic"], c["dynamic"], a["rangeofmotion"], c["rangeofmotion"], b["dynamic"], d["dynamic"], b["rangeofmotion"], d["rangeofmotion"]), flush=True)
    del x, g, j
    torch.cuda.empty_cache()
