"""Step time of C3 batches of 64 .. 512 problems (one structure, distinct x) launched back to back: what a chunked
evaluation of the 8192-problem batch would cost per chunk."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs

model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
xs = perturbed_inputs(S, model, 256, 0)
for B in [int(a) for a in sys.argv[1:]] or [64, 96, 128, 160, 192, 224, 256, 320, 512]:
    batch = ta.Batch([S], [0] * B, device=0)
    x = torch.from_numpy(np.concatenate([xs[i % 256] for i in range(B)])).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(30):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print("B=%4d  %.1f us/step  %.2f M cb/s  -> 8192 problems in %.3f ms" % (B, us, B / us, 8192 / B * us / 1e3), flush=True)
