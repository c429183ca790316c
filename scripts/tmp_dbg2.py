import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import towr_amd as ta
from tests.common import Case
cases = [Case("anymal", "gap", ta.gait_combo(4, 1, 2.0), constraint_sets=127),
         Case("anymal", "stairs", ta.gait_combo(4, 2, 1.7), constraint_sets=63),
         Case("anymal", "flat", ta.gait_combo(4, 0, 2.2), constraint_sets=255, base_z_init=0.5)]
order = [0, 1, 2, 0, 1]
batch = ta.Batch([c.S for c in cases], order, device=0)
xs = np.concatenate([cases[s].x_wild(70 + p) for p, s in enumerate(order)])
g, j = batch.eval_host(xs)
px, pg, pj = batch.host_buffers()
px[:] = xs
for flags in (3, 1, 2):
    pg[:] = np.nan; pj[:] = np.nan
    batch.eval_host_pinned(flags)
    for p, s in enumerate(order):
        a, b = batch.g_off[p], batch.g_off[p + 1]
        dg = np.nonzero(~((pg[a:b] == g[a:b]) | (np.isnan(pg[a:b]) & (flags == 2))))[0]
        a2, b2 = batch.jac_off[p], batch.jac_off[p + 1]
        dj = np.nonzero(pj[a2:b2] != j[a2:b2])[0] if flags & 2 else []
        if len(dg) or len(dj):
            S = cases[s].S
            names = [(c["name"], c["offset"], c["size"]) for c in S.con_sets]
            print("flags", flags, "problem", p, "struct", s, "g diffs", len(dg), dg[:10], "nan?", np.isnan(pg[a:b][dg][:5]), "jac diffs", len(dj))
            for nm, off, sz in names:
                k = ((dg >= off) & (dg < off + sz)).sum()
                if k: print("    set", nm, k)
g2, _ = batch.eval_host(xs, 1)
print("pageable values-only equal:", np.array_equal(g2, g))
