"""One-off stress: many random structures per leg count, fixed timings only in one batch (fused launch) and mixed
(separate launches), both flag subsets, against the oracle."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from tests.common import random_case, assert_parity

def split(batch, g, j, p):
    return g[batch.g_off[p]:batch.g_off[p + 1]], j[batch.jac_off[p]:batch.jac_off[p + 1]]

cases = [random_case(5000 + s) for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 240)]
bad = 0
for n_ee in (1, 2, 4):
    for timings in (False, True):
        group = [c for c in cases if c.S.n_ee == n_ee and bool(c.S.params.constraint_sets & 64) == timings]
        if not group:
            continue
        # odd and even counts, repeated structures
        order = list(range(len(group))) + [0, len(group) // 2]
        batch = ta.Batch([c.S for c in group], order, device=0)
        xs = [group[s].x_wild(70 + i) for i, s in enumerate(order)]
        g, j = batch.eval_host(np.concatenate(xs))
        for p, s in enumerate(order):
            rg, _, _, rj = group[s].P.eval(xs[p])
            gd, jd = split(batch, g, j, p)
            try:
                assert_parity(group[s].S, gd, jd, rg, rj, "n_ee %d problem %d" % (n_ee, p), x=xs[p])
            except AssertionError as e:
                bad += 1
                print("MISMATCH n_ee", n_ee, "timings", timings, "problem", p, str(e)[:300], flush=True)
        print("n_ee %d timings %s: %d problems checked" % (n_ee, timings, len(order)), flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
