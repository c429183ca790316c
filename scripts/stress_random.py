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
        # round 5: the values-only kernels (their own launch path) and the scoring kernel on the same batch: g of the
        # values-only evaluation against g of values + Jacobian (to rounding), scores against numpy on that g and the bounds
        x_d = torch.from_numpy(np.concatenate(xs)).cuda()
        g_d = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
        sc_d = torch.empty((len(order), 16), dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        batch.eval_device(x_d.data_ptr(), g_d.data_ptr(), 0, ta.EVAL_VALUES, st)
        batch.score_device(g_d.data_ptr(), sc_d.data_ptr(), st)
        torch.cuda.synchronize()
        gv, sc = g_d.cpu().numpy(), sc_d.cpu().numpy()
        for p, s in enumerate(order):
            S = group[s].S
            a, b = batch.g_off[p], batch.g_off[p + 1]
            scale = max(1.0, np.abs(g[a:b]).max()) if b > a else 1.0
            if b > a and not np.abs(gv[a:b] - g[a:b]).max() <= 1e-12 * scale:
                bad += 1
                print("VALUES-ONLY MISMATCH n_ee", n_ee, "problem", p, np.abs(gv[a:b] - g[a:b]).max(), flush=True)
            lo, up = S.bounds()
            viol = np.maximum(np.maximum(lo - gv[a:b], gv[a:b] - up), 0.0)
            want = np.zeros((8, 2))
            for cs in S.con_sets:
                fam = [i for i, f in enumerate(ta.FAMILIES) if cs["name"].startswith(f)][0]
                v = viol[cs["offset"]:cs["offset"] + cs["size"]]
                if v.size:
                    want[fam, 0] = max(want[fam, 0], v.max())
                    want[fam, 1] += v.sum()
            got = sc[p].reshape(8, 2)
            if not (np.array_equal(got[:, 0], want[:, 0]) and np.allclose(got[:, 1], want[:, 1], rtol=1e-12, atol=0)):
                bad += 1
                print("SCORE MISMATCH n_ee", n_ee, "problem", p, got.tolist(), want.tolist(), flush=True)
        print("n_ee %d timings %s: %d problems checked" % (n_ee, timings, len(order)), flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
