"""Is the slow / fast state of an output allocation (first_leg_probe*.py) tied to the REGULAR problem stride of the C3 batch (8192
identical structures: every problem's Jacobian 823 168 B after the previous one)?  The same eight allocations evaluated with (a) the
uniform batch and (b) a batch that alternates K = 200 and K = 199 structures (irregular strides), per-byte rates compared."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup, PLACEMENT_BALLAST_GB

dev = torch.device("cuda", 0)
model = ta.model_preset("anymal", "flat")
_, _, S200 = build_case(ta, model, K=200)
_, _, S199 = build_case(ta, model, K=199)
B = 8192
uni = ta.Batch([S200], [0] * B, device=0)
mix = ta.Batch([S200, S199], [i % 2 for i in range(B)], device=0)
st = torch.cuda.current_stream().cuda_stream
device_power_warmup(torch, dev, 0.5)


def xs(batch, structs, order):
    base = {id(s): perturbed_inputs(s, model, 16, 0) for s in structs}
    return torch.from_numpy(np.concatenate([base[id(structs[o])][i % 16] for i, o in enumerate(order)])).to(dev)


x_uni, x_mix = xs(uni, [S200], [0] * B), xs(mix, [S200, S199], [i % 2 for i in range(B)])


def timed(batch, x, g, j):
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20


for i, gb in enumerate(PLACEMENT_BALLAST_GB):
    ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb else None
    g = torch.empty(int(uni.g_off[-1]), dtype=torch.float64, device=dev)
    j = torch.empty(int(uni.jac_off[-1]), dtype=torch.float64, device=dev)
    del ballast
    tu, tm = timed(uni, x_uni, g, j), timed(mix, x_mix, g, j)
    print("allocation %d: uniform %.3f ms = %.2f TB/s     alternating K = 200 / 199: %.3f ms = %.2f TB/s" % (
        i, tu * 1e3, uni.algorithmic_bytes / tu / 1e12, tm * 1e3, mix.algorithmic_bytes / tm / 1e12), flush=True)
    del g, j
    torch.cuda.empty_cache()
