"""Kernel times of the default workload with values+Jacobian, Jacobian only, values only (per-kernel HIP events)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs

model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
xs = perturbed_inputs(S, model, 256, 0)
x = torch.from_numpy(np.concatenate([xs[i % 256] for i in range(B)])).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for name, fl in (("both", ta.EVAL_BOTH), ("jacobian", ta.EVAL_JACOBIAN), ("values", ta.EVAL_VALUES)):
        for _ in range(3):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), fl, st)
        batch.profile_begin(20)
        for _ in range(20):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), fl, st)
        torch.cuda.synchronize()
        ms, n = batch.profile_end()
        print("%-9s dyn %.3f  rom %.3f  nodes %.3f ms" % (name, ms["dynamic"], ms["rangeofmotion"], ms["nodes"]), flush=True)
