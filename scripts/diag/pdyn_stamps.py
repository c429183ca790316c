import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs
model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model, constraint_sets=127)
B = 2048
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
x = torch.from_numpy(np.tile(base, (B // 256, 1)).reshape(-1)).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
jac = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, st)
torch.cuda.synchronize()
out = np.zeros(1024 * 8, dtype=np.uint64)
L = ta.lib()
L.twr_debug_pdyn_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.twr_debug_pdyn_stamps(out.ctypes.data_as(C.c_void_p), out.size) == 0
a = out.reshape(-1, 8).astype(np.float64)
a = a[a[:, 7] > 0]
per = a[:, :7] / a[:, 7:8]
names = ["top: wait x, issue P R", "clear", "math", "wait P", "puts", "wait R, issue X", "stream"]
tot = per.sum(axis=1).mean()
print("workgroups %d, runs per workgroup %.1f, memtime ticks per run %.0f" % (len(a), a[:, 7].mean(), tot))
for n, v in zip(names, per.mean(axis=0)):
    print("  %-20s %9.0f  %5.1f %%" % (n, v, 100 * v / tot))
