"""Small-batch step time of the ragged sweep (no per-kernel events): 32 -> 512 candidates on one GPU."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
for B in (32, 64, 128, 256, 512):
    cands = sweep.enumerate_candidates(B)
    structs = [sweep.candidate_structure(model, c) for c in cands]
    batch = ta.Batch(structs, list(range(B)), device=0)
    xh = np.concatenate([perturbed_inputs(s, model, 1, i)[0] for i, s in enumerate(structs)])
    x = torch.from_numpy(xh).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(30):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print("B=%4d  %.1f us/step  %.3e callbacks/s" % (B, us, B / us * 1e6), flush=True)
