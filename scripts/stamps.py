#!/usr/bin/env python3
"""Per-phase cycle breakdown of dyn_kernel from the s_memtime-stamped diagnostic build (make -C towr_amd/csrc ablate ->
libtowr_amd_stamps.so).  Every stamp drains the wave's LDS/scalar queue (s_memtime returns through lgkmcnt), so the build
is slower than the product and a phase is charged with the latency of what it issued; the split is what matters."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TWR_AMD_LIB"] = os.path.join(ROOT, "towr_amd", "libtowr_amd_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import towr_amd as ta  # noqa: E402
from bench import build_case, perturbed_inputs  # noqa: E402

model = ta.model_preset("anymal", "flat")
sched, params, S = build_case(ta, model)
B = 8192
batch = ta.Batch([S], [0] * B, device=0)
base = perturbed_inputs(S, model, 256, 0)
x = torch.from_numpy(np.tile(base, (B // 256, 1)).reshape(-1)).cuda()
g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
jac = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, st)
torch.cuda.synchronize()
out = np.zeros(2048 * 8, dtype=np.uint64)
L = ta.lib()
L.twr_debug_dyn_stamps.argtypes = [C.c_void_p, C.c_int]
rc = L.twr_debug_dyn_stamps(out.ctypes.data_as(C.c_void_p), out.size)
assert rc == 0, rc
a = out.reshape(-1, 8).astype(np.float64)
a = a[a[:, 6] > 0]
per = a[:, :6] / a[:, 6:7]
names = ["loop/work item", "front", "copy-out", "back (+ put-record wait)", "stage x (+ gather wait)", "issue loads"]
tot = per.sum(axis=1).mean()
print("workgroups %d, slices per workgroup %.1f, cycles per slice %.0f" % (len(a), a[:, 6].mean(), tot))
for n, v in zip(names, per.mean(axis=0)):
    print("  %-28s %8.0f cycles  %5.1f %%" % (n, v, 100 * v / tot))
