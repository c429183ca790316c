"""Follow-up of first_leg_probe.py: is only the FIRST allocation of the process slow, or does it depend on what was allocated
before / how?  Each line: fresh x / g / jac buffers for the same C3 batch, 20 timed steps."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda", 0)
if mode == "prealloc":   # a dummy allocate / free cycle of the output size BEFORE anything else
    d = torch.empty(860_000_000, dtype=torch.float64, device=dev)
    d.fill_(0.0)
    torch.cuda.synchronize()
    del d
    torch.cuda.empty_cache()
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
xh = np.tile(base, (B // 256, 1)).reshape(-1)
st = torch.cuda.current_stream().cuda_stream


def buffers():
    return (torch.from_numpy(xh).to(dev), torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev),
            torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev))


def region(buf, what):
    x, g, j = buf
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    batch.profile_begin(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    k, _ = batch.profile_end()
    print("%-40s %.3f ms/step  dyn %.3f rom %.3f   jac at 0x%x" % (what, ms, k["dynamic"], k["rangeofmotion"], j.data_ptr()), flush=True)


device_power_warmup(torch, dev, 0.5)
for i in range(4):
    b = buffers()
    region(b, "%s: allocation %d" % (mode, i))
    del b
    torch.cuda.empty_cache()
b = buffers()
b[2].zero_()
region(b, "%s: allocation 4, zeroed first" % mode)
