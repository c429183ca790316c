"""Step time of sweeps LARGER than the enumeration (its 1024 candidates repeated r times as separately built structures:
every copy has its own times / rom tables, the layout tables merge by content), event-free, for the store-policy decision
at sizes whose tables exceed the Infinity Cache.  Usage: python scripts/big_sweep.py 1 2 4 8   (tuning library: TWR_STREAM_NT=0|1)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
for r in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    cands = sweep.enumerate_candidates(1024) * r
    structs = sweep.candidate_structures(model, cands)
    B = len(structs)
    batch = ta.Batch(structs, list(range(B)), device=0)
    tb = batch.table_bytes()
    xh = np.concatenate([perturbed_inputs(s, model, 1, i % 1024)[0] for i, s in enumerate(structs)])
    x = torch.from_numpy(xh).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print("B=%5d  nt=%d  tables %.0f MB resident (dyn layout %.0f -> %.0f MB)  %.1f us/step  %.2f M cb/s  %.2f TB/s"
          % (B, batch.streaming_stores(), tb["resident"] / 1e6, tb["dyn_layout"] / 1e6, tb["dyn_layout_distinct"] / 1e6, us, B / us,
             batch.algorithmic_bytes / us / 1e6), flush=True)
    del batch, x, g, j
