"""A sweep at towr's DEFAULT grids (dt_constraint_dynamic 0.1 s, dt_constraint_range_of_motion 0.08 s, parameters.cc:49-50)
instead of the BASELINE configurations' K = 200: 1024 Stairs candidates, values only and values + Jacobian, event-free."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from towr_amd import sweep
from bench import perturbed_inputs

model = ta.model_preset("anymal", "stairs")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cands = sweep.enumerate_candidates(B)
structs = ta.Structure.create_many(model, [ta.gait_combo(model.n_ee, combo, T, scale) for combo, T, scale in cands], [ta.params_default() for _ in cands])
S0 = structs[0]
print("first candidate: n=%d m=%d nnz=%d k_dynamic=%d k_rom=%d items=%s" % (S0.n, S0.m, S0.nnz, S0.k_dynamic, S0.k_rom, S0.values_items()))
batch = ta.Batch(structs, list(range(B)), device=0)
xh = np.concatenate([perturbed_inputs(s, model, 1, i)[0] for i, s in enumerate(structs)])
x = torch.from_numpy(xh).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def timed(flags, jp, n=300):
    for _ in range(20):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jp, flags, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jp, flags, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


tv, tb = timed(ta.EVAL_VALUES, 0), timed(ta.EVAL_BOTH, j.data_ptr())
bytes_both = 8 * (int(batch.x_off[-1]) + int(batch.g_off[-1]) + int(batch.jac_off[-1]))
print("B=%d default grids: values only %.1f us, values + Jacobian %.1f us (%.2f TB/s on %.1f MB)" % (B, tv, tb, bytes_both / tb / 1e6, bytes_both / 1e6))
