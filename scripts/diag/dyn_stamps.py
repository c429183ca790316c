"""Per-phase split of dyn_kernel's loop from the stamped diagnostic library (scripts/diag/stamp_dyn.py):
C3, 8192 problems, three separate launches (per-kernel events on, so that dyn_kernel itself runs, not the fused launch)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs
# workload: c3 (8192 problems, one structure) | a1024 (1024 problems, one structure) | b1024 (1024 separately built copies of that
# structure) | sweep (the 1024 enumerated C5 candidates): what a sweep's own tables cost, phase by phase
WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
if WL == "sweep":
    from towr_amd import sweep
    model = ta.model_preset("anymal", "stairs")
    structs = sweep.candidate_structures(model, sweep.enumerate_candidates(1024))
    B = 1024
    batch = ta.Batch(structs, list(range(B)), device=0)
    x = torch.from_numpy(np.concatenate([perturbed_inputs(s_, model, 1, i)[0] for i, s_ in enumerate(structs)])).cuda()
else:
    model = ta.model_preset("anymal", "flat")
    sched, params, S = build_case(ta, model)
    B = 8192 if WL == "c3" else 1024
    if WL == "b1024":
        structs = ta.Structure.create_many(model, [sched] * B, [params] * B, 0)
        batch = ta.Batch(structs, list(range(B)), device=0)
    else:
        batch = ta.Batch([S], [0] * B, device=0)
    base = perturbed_inputs(S, model, 256, 0)
    x = torch.from_numpy(np.tile(base, (B // 256, 1)).reshape(-1)).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
jac = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
batch.profile_begin(5)
for _ in range(5):
    batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, st)
torch.cuda.synchronize()
ms, _ = batch.profile_end()
out = np.zeros(1024 * 8, dtype=np.uint64)
L = ta.lib()
L.twr_debug_dyn_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.twr_debug_dyn_stamps(out.ctypes.data_as(C.c_void_p), out.size) == 0
a = out.reshape(-1, 8).astype(np.float64)
a = a[a[:, 7] > 0]
per = a[:, :5] / a[:, 7:8]
names = ["F front (waits for last S's loads, LDS reads of xs, spline points, sincos)", "P issue codes + next front records",
         "O copy-out of the previous image", "B back (tile + base blocks -> image)", "S stage x, issue selector / gather / map"]
tot = per.sum(axis=1).mean()
print(WL, "dyn_kernel (stamped build) %.3f ms; workgroups %d, slices per workgroup %.1f, memtime ticks per slice %.0f"
      % (ms["dynamic"], len(a), a[:, 7].mean(), tot))
for n, v in zip(names, per.mean(axis=0)):
    print("  %-80s %9.0f  %5.1f %%" % (n, v, 100 * v / tot))
