"""For a rocprofv3 --pmc pass: the C3 batch evaluated ten times into each of several Jacobian allocations of one process (the slow /
fast state of an allocation, first_leg_probe*.py).  scripts/alloc_states_summary.py groups the dispatches by allocation."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs, device_power_warmup, PLACEMENT_BALLAST_GB

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
batch.profile_begin(1000)   # (separate launches, as in the headline's timed region)
for i, gb in enumerate(PLACEMENT_BALLAST_GB):
    ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb else None
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    del ballast
    for _ in range(10):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    del j
    torch.cuda.empty_cache()
