import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import towr_amd as ta
from common import Case, k_params
case = Case("anymal", "flat", ta.gait_combo(4, 1, 2.0), constraint_sets=127, **k_params(2.0, 200))
S = case.S
B, nb = 512, 16
batch = ta.Batch([S], [0] * B, device=0)
base = np.stack([case.x_perturbed(40 + i) for i in range(nb)])
x = torch.from_numpy(np.tile(base, (B // nb, 1)).reshape(-1)).cuda()
g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
torch.cuda.synchronize()
G, J = g.view(B, S.m).cpu().numpy(), j.view(B, S.nnz).cpu().numpy()
print("finite", np.isfinite(G).all(), np.isfinite(J).all())
sets = S.con_sets
for p in range(B):
    ref = p % nb
    dg = np.nonzero(G[p] != G[ref])[0]
    dj = np.nonzero(J[p] != J[ref])[0]
    if len(dg) or len(dj):
        print("problem", p, "g diffs", len(dg), dg[:8], "jac diffs", len(dj), dj[:12])
        for idx in dj[:6]:
            row = np.searchsorted(S.row_ptr, idx, side="right") - 1
            print("   jac idx", idx, "row", row, "col", S.col_idx[idx], J[p][idx], J[ref][idx])
        break
nd = sum(1 for p in range(B) if (J[p] != J[p % nb]).any() or (G[p] != G[p % nb]).any())
print("problems differing:", nd, "of", B)
rg, _, _, rj = case.P.eval(base[0])
print("problem 0 vs oracle: g", np.abs(G[0] - rg).max(), "jac", np.abs(J[0] - rj).max())
for s in sets: print(s["name"], s["offset"], s["size"], s["nnz_offset"], s["nnz"])
