"""Host-buffer callback (twr_batch_eval_host on the batch's page-locked buffers) for B = 1 .. 64 problems; run with
TWR_HOST_ZERO_COPY=0 for the copy path (H2D, eval into HBM, two D2H copies) instead of kernel stores into host memory."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import towr_amd as ta
from bench import build_case, perturbed_inputs
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
for B in (1, 2, 4, 8, 16, 32, 64):
    batch = ta.Batch([S], [0] * B, device=0)
    px, pg, pj = batch.host_buffers()
    xs = perturbed_inputs(S, model, B, 0)
    px[:] = np.concatenate(xs)
    for _ in range(20): batch.eval_host_pinned()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): batch.eval_host_pinned()
    print("B=%d %.1f us" % (B, (time.perf_counter() - t0) / n * 1e6), flush=True)
