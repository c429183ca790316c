"""C3, 8192 problems, ONE process, ONE set of buffers: the three launches (per-kernel events on: the library then never fuses) against
the fused launch (events off; TUNING build with TWR_FUSED_MAX_ROM raised and TWR_FUSED_GROM / TWR_FUSED_GDYN set).  Same allocation
for both, so the slow / fast state of an allocation cancels."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

dev = torch.device("cuda", 0)
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
xh = np.tile(base, (B // 256, 1)).reshape(-1)
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, dev, 0.5)
x = torch.from_numpy(xh).to(dev)
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)


def run(n, events):
    if events:
        batch.profile_begin(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    if events:
        batch.profile_end()
    return ms


run(10, True); run(10, False)
sep, fus = [], []
for _ in range(4):
    sep.append(run(20, True))
    fus.append(run(20, False))
print("three launches (with events) %s   fused %s   fused / separate %.3f" % (["%.3f" % v for v in sep], ["%.3f" % v for v in fus], min(fus) / min(sep)), flush=True)
