"""Why is the FIRST timed leg of bench.py 5-10 % slower than the same kernels a few seconds later in the same process
(rom_kernel 0.94 vs 0.85 ms on some boxes)?  Clocks (time under load) or placement (which memory the buffers got)?
Same C3 batch, 20-step regions: (a) fresh buffers right after set-up, (b) the same buffers after 6 s of load, (c) new buffers
allocated then, (d) the first buffers again."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
xh = np.tile(base, (B // 256, 1)).reshape(-1)
dev = torch.device("cuda", 0)
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
    print("%-52s %.3f ms/step  dyn %.3f rom %.3f node %.3f   jac at 0x%x" % (what, ms, k["dynamic"], k["rangeofmotion"], k["nodes"], j.data_ptr()), flush=True)


device_power_warmup(torch, dev, 0.5)
a = buffers()
region(a, "(a) fresh buffers, right after set-up")
region(a, "(a) again")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 6.0:
    for _ in range(50):
        batch.eval_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
region(a, "(b) the same buffers after 6 s of load")
c = buffers()
region(c, "(c) new buffers (the first ones still allocated)")
region(a, "(d) the first buffers again")
del a
torch.cuda.empty_cache()
e = buffers()
region(e, "(e) new buffers after freeing the first (empty_cache)")
time.sleep(3.0)
region(e, "(f) the same after 3 s idle")
