"""C3, 8192 problems: the Jacobian buffer as a WINDOW of one large arena allocation -- the step time for windows every few GB
(which pieces of the device's memory take the kernels' store streams faster).
usage: arena_probe.py [arena GB] [window step GB]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup

dev = torch.device("cuda", 0)
arena_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
step_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
x = torch.from_numpy(np.tile(base, (B // 256, 1)).reshape(-1)).to(dev)
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, dev, 0.5)
arena = torch.empty(int(arena_gb * (1 << 27)), dtype=torch.float64, device=dev)
n = int(batch.jac_off[-1])
print("arena %.0f GB @ %#x, window %.2f GB" % (arena_gb, arena.data_ptr(), n * 8 / 2 ** 30), flush=True)
off, rows = 0, []
while off + n <= arena.numel():
    j = arena[off:off + n]
    for _ in range(3):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    rows.append((off * 8 / 2 ** 30, (time.perf_counter() - t0) / 8 * 1e3))
    off += int(step_gb * (1 << 27))
print(" ".join("%.0f:%.3f" % r for r in rows), flush=True)
best = min(rows, key=lambda r: r[1])
print("fastest window at %.0f GB: %.3f ms/step = %.3f M callbacks/s; slowest %.3f ms" % (best[0], best[1], B / best[1] / 1e3, max(r[1] for r in rows)))
