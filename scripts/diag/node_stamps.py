"""Per-phase split of node_chunk_kernel's loop from the stamped diagnostic library (scripts/diag/stamp_node_chunk.py):
C3 with towr's whole default constraint list, 8192 problems."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model, constraint_sets=63)
B = 8192
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
out = np.zeros(4 * 256 * 8, dtype=np.uint64)
L = ta.lib()
L.twr_debug_node_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.twr_debug_node_stamps(out.ctypes.data_as(C.c_void_p), out.size) == 0
a = out.reshape(4, 256, 8).astype(np.float64)
print("node_chunk_kernel (stamped build) %.3f ms" % ms["nodes"])
names = ["issue work item + record + x loads (waits for the work item three ahead)", "compute -> LDS (waits for x, records)", "copy-out + g stores (issue)", "rotate (waits for the in-flight records / x)"]
for fam, fn in enumerate(("terrain", "force", "splineacc", "swing")):
    f = a[fam]
    f = f[f[:, 7] > 0]
    if not len(f):
        continue
    per = f[:, :4] / f[:, 7:8]
    print("%-9s chunks per wave %.1f, ticks per chunk %.0f:" % (fn, f[:, 7].mean(), per.sum(axis=1).mean()),
          "  ".join("%s %.0f" % (n.split(" ")[0], v) for n, v in zip(names, per.mean(axis=0))))
