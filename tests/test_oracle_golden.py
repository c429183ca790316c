"""Pins the CPU oracle (oracle/towr_oracle.cc).

The reference ships no golden vectors for this path (its two gtest files are empty stubs) and cannot
be built here, so the oracle is pinned by (1) fixtures from an independent 40-digit mpmath
implementation that differentiates g numerically (oracle/mp_ref.py -> tests/golden/mp_*.npz),
(2) hand known-answers derived from the cited reference lines, (3) finite
differences of the oracle's own g (what Ipopt's derivative_test would do, hopper_example.cc:86).
"""
import glob
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mp_*.npz")))


def load_fixture(path):
    d = np.load(path)
    pd, o = [], 0
    for k in d["n_phases"]:
        pd.append(d["phase_durations"][o:o + k])
        o += k
    sets = int(d["constraint_sets"]) if "constraint_sets" in d.files else ob.SETS_HOT_PATH
    # (fixtures at BASELINE sizes carry their discretisation; the older ones use the reference defaults 0.1 / 0.08)
    dts = dict(dt_dynamic=float(d["dt_dynamic"]), dt_rom=float(d["dt_rom"])) if "dt_dynamic" in d.files else {}
    # (the fpowr fixture runs on the `Grid` terrain: it carries its grid_map elevation layer)
    gm = dict(grid_map=(d["grid_elevation"], float(d["grid_resolution"]), tuple(d["grid_position"]))) if "grid_elevation" in d.files else {}
    P = ob.OracleProblem(str(d["robot"]), str(d["terrain"]), pd, list(d["contact_at_start"]), constraint_sets=sets,
                         base_z_init=0.6, **dts, **gm)
    return d, P


def quirk_mask(P, terrain, x, rp, ci):
    """Force rows x stance-foothold columns on a terrain with curvature: the reference's
    GetDerivativeOfNormalizedBasisWrt (height_map.cc:80-91) is a component-wise product, not the
    chain rule, so these entries are not derivatives of g."""
    mask = np.zeros(P.nnz, dtype=bool)
    if terrain != "gap":
        return mask
    motion_cols = {}
    off = 0
    for name, size in P.var_sets:
        if name.startswith("ee-motion_"):
            motion_cols[int(name.split("_")[1])] = (off, off + size)
        off += size
    r0 = 0
    for name, rows in P.con_sets:
        if name.startswith("force-"):
            a, b = motion_cols[int(name.split("_")[-1])]
            for r in range(r0, r0 + rows):
                for k in range(rp[r], rp[r + 1]):
                    if a <= ci[k] < b:
                        mask[k] = True
        r0 += rows
    return mask


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[3:-4] for p in GOLDEN])
def test_oracle_matches_independent_mpmath_golden(path):
    d, P = load_fixture(path)
    x = d["x"]
    assert P.n == x.size and P.m == d["g"].size
    g, rp, ci, va = P.eval(x)
    # constraint values: 1e-13 of the set scale (mp value rounded to double)
    assert np.abs(g - d["g"]).max() <= 1e-13 * np.abs(d["g"]).max()
    ref = {(int(r), int(c)): v for r, c, v in zip(d["jac_row"], d["jac_col"], d["jac_val"])}
    rows = np.repeat(np.arange(P.m), np.diff(rp))
    have = set(zip(rows.tolist(), ci.tolist()))
    # every true non-zero lies inside the oracle's (x-independent) pattern
    assert all(k in have for k in ref)
    refv = np.array([ref.get((r, c), 0.0) for r, c in zip(rows.tolist(), ci.tolist())])
    rowscale = np.zeros(P.m)
    np.maximum.at(rowscale, rows, np.abs(refv))
    tol = 1e-9 * np.abs(refv) + 1e-12 * rowscale[rows]
    bad = np.abs(va - refv) > tol
    qm = quirk_mask(P, str(d["terrain"]), x, rp, ci)
    assert not (bad & ~qm).any(), "%d analytic Jacobian entries disagree with numerical differentiation" % (bad & ~qm).sum()
    if str(d["terrain"]) == "gap":
        assert (bad & qm).any(), "fixture should exercise the terrain-basis quirk (footholds inside the gap)"


def test_terrain_basis_derivative_quirk_known_answer():
    """height_map.cc:80-91,141-148 evaluated by hand at a point inside the Gap parabola."""
    x, y = 1.1, 0.3
    w, h, gs = 0.5, 1.5, 1.0
    xc = gs + w / 2
    a, b = 4 * h / (w * w), -(8 * h * xc) / (w * w)
    hx, hxx = 2 * a * x + b, 2 * a
    for which, v, dv in ((0, np.array([-hx, 0.0, 1.0]), np.array([-hxx, 0.0, 0.0])),
                         (1, np.array([1.0, 0.0, hx]), np.array([0.0, 0.0, hxx]))):
        nrm = np.linalg.norm(v)
        unit = np.array([1.0, 0.0, 0.0])
        outer = (1 / (nrm * nrm)) * (nrm * unit - v[0] * (v / nrm))
        expect = outer * dv  # component-wise, exactly as the reference
        got = ob.terrain_dbasis("gap", which, 0, x, y)
        assert np.allclose(got, expect, rtol=1e-14, atol=0)
        # ... and it is NOT the derivative of the normalised vector (documented quirk)
        eps = 1e-6

        def nb(xx):
            hx2 = 2 * a * xx + b
            vv = np.array([-hx2, 0.0, 1.0]) if which == 0 else np.array([1.0, 0.0, hx2])
            return vv / np.linalg.norm(vv)

        true = (nb(x + eps) - nb(x - eps)) / (2 * eps)
        assert np.abs(true - expect).max() > 1e-3
    assert np.array_equal(ob.terrain_dbasis("gap", 2, 0, x, y), np.zeros(3))
    assert np.array_equal(ob.terrain_dbasis("gap", 0, 1, x, y), np.zeros(3))


def test_hermite_weights_known_answers():
    """polynomial.cc:140-234 at the segment ends."""
    T = 0.37
    w0, wT = ob.hermite_weights(0.0, T), ob.hermite_weights(T, T)
    assert np.allclose(w0[0], [1, 0, 0, 0], atol=1e-15) and np.allclose(wT[0], [0, 0, 1, 0], atol=1e-14)
    assert abs(w0[1][1] - 1) < 1e-15 and abs(wT[1][3] - 1) < 1e-13
    assert np.allclose(w0[2], [-6 / T**2, -4 / T, 6 / T**2, -2 / T], rtol=1e-14)
    # partition of unity of the position weights on p0,p1 and exactness on cubics
    t = 0.123
    w = ob.hermite_weights(t, T)
    assert abs(w[0][0] + w[0][2] - 1) < 1e-15
    c = np.array([0.3, -1.2, 0.7, 2.1])  # p(t) = c0 + c1 t + c2 t^2 + c3 t^3
    p = lambda s: c @ [1, s, s * s, s**3]
    v = lambda s: c @ [0, 1, 2 * s, 3 * s * s]
    acc = lambda s: c @ [0, 0, 2, 6 * s]
    nodes = np.array([p(0), v(0), p(T), v(T)])
    assert abs(w[0] @ nodes - p(t)) < 1e-14 and abs(w[1] @ nodes - v(t)) < 1e-13 and abs(w[2] @ nodes - acc(t)) < 1e-11


def test_terrain_known_answers():
    """height_map_examples.{h,cc}: heights, inclusive range tests, Stairs has zero slope."""
    assert ob.terrain_height("flat", 3.0, -2.0) == 0.0
    assert ob.terrain_height("stairs", 0.99, 0) == 0.0 and ob.terrain_height("stairs", 1.0, 0) == 0.2
    assert ob.terrain_height("stairs", 1.4, 0) == 0.4 and ob.terrain_height("stairs", 2.4, 0) == 0.0
    assert np.array_equal(ob.terrain_basis("stairs", 0, 1.2, 0.0), [0, 0, 1])
    assert ob.terrain_height("gap", 1.0, 0) == pytest.approx(0.0, abs=1e-12)
    assert ob.terrain_height("gap", 1.25, 0) == pytest.approx(-1.5, abs=1e-12)
    assert ob.terrain_height("gap", 1.5000001, 0) == 0.0
    assert ob.terrain_height("block", 0.7 + 0.03, 0) == 0.5 and ob.terrain_height("block", 0.715, 0) == pytest.approx(0.25)
    assert ob.terrain_height("slope", 1.5, 0) == pytest.approx(0.35) and ob.terrain_height("slope", 2.5, 0) == pytest.approx(0.35)
    assert ob.terrain_height("chimney", 1.2, 0.7) == pytest.approx(0.6) and ob.terrain_height("chimney", 0.9, 0.7) == 0.0
    assert ob.terrain_height("chimney_lr", 1.0, 0.2) == pytest.approx(-0.6) and ob.terrain_height("chimney_lr", 2.0, 0.2) == pytest.approx(-1.4)
    n = ob.terrain_basis("slope", 0, 1.5, 0.0)
    assert np.allclose(n, np.array([-0.7, 0, 1]) / np.hypot(0.7, 1), rtol=1e-15)


def test_csv_grid_terrain_known_answers():
    """HeightMapFromCSV (terrain/height_map_from_csv.h:29-109): height of the 0.17 m cell under (x, y), 0 outside
    the grid; the slope is a step to the next / previous cell smeared over eps = cell/50 on the lower side."""
    grid = np.array([[0.0, 0.1, 0.1], [0.2, 0.05, 0.3]])   # grid[y_cell, x_cell]
    P = ob.OracleProblem("monoped", "csv", [[0.4, 0.2, 0.4]], [1], grid=grid)
    off = 0
    for name, size in P.var_sets:
        if name == "ee-motion_0":
            m = off
        off += size
    res, eps = 0.17, 0.17 / 50

    def terrain_row(px, py, node=1):
        x = np.zeros(P.n)
        # ee-motion_0: stance (px py pz) | swing node px vx py vy pz | stance ...; terrain row 0 is node 1 = the
        # end node of the first stance phase = the same variables as node 0
        x[m:m + 3] = [px, py, 0.7]
        g, rp, ci, va = P.eval(x)
        return g[0], va[rp[0]:rp[1]]

    g0, j0 = terrain_row(0.5 * res, 0.5 * res)              # cell (0,0), far from any edge
    assert g0 == 0.7 - 0.0 and np.array_equal(j0, [-0.0, -0.0, 1.0])
    g0, j0 = terrain_row(1.5 * res, 1.5 * res)              # cell x=1, y=1 -> 0.05
    assert g0 == pytest.approx(0.7 - 0.05, abs=1e-15)
    g0, j0 = terrain_row(res - 0.5 * eps, 0.5 * res)        # just below the x edge to a higher cell: slope +0.1/eps
    assert g0 == 0.7 and np.allclose(j0, [-0.1 / eps, 0.0, 1.0], rtol=1e-14)
    g0, j0 = terrain_row(res + 0.5 * eps, 1.5 * res)        # cell (y=1,x=1)=0.05 just after a drop from 0.2: slope -0.15/eps
    assert np.allclose(j0, [0.15 / eps, 0.0, 1.0], rtol=1e-13)
    g0, j0 = terrain_row(0.5 * res, res - 0.5 * eps)        # just below the y edge, next row higher (0.2): dh/dy = 0.2/eps
    assert np.allclose(j0, [0.0, -0.2 / eps, 1.0], rtol=1e-14)
    g0, j0 = terrain_row(3.5 * res, 0.5 * res)              # outside the grid: height 0, no slope
    assert g0 == 0.7 and np.array_equal(j0, [-0.0, -0.0, 1.0])
    g0, j0 = terrain_row(-0.3 * res, 0.5 * res)             # (-1, 0) * res truncates to cell 0, like static_cast<size_t>
    assert g0 == 0.7 - 0.0
    g0, j0 = terrain_row(-1.3 * res, 0.5 * res)             # negative cell: outside
    assert g0 == 0.7


def _anymal_static():
    pd, con = [[1.0]] * 4, [1] * 4          # one stance phase per foot: the robot just stands
    P = ob.OracleProblem("anymal", "flat", pd, con)
    ee = [[0.34, 0.19, 0], [0.34, -0.19, 0], [-0.34, 0.19, 0], [-0.34, -0.19, 0]]
    x = P.initial_guess([0, 0, 0.42], [0, 0, 0], [0, 0, 0.42], [0, 0, 0], ee)  # goal == start: standing still
    return P, x


def test_static_stance_known_answers():
    """standing still on flat ground with m g/4 per leg."""
    P, x = _anymal_static()
    g, rp, ci, va = P.eval(x)
    sets, r0 = {}, 0
    for name, rows in P.con_sets:
        sets[name] = (r0, r0 + rows)
        r0 += rows
    a, b = sets["dynamic"]
    assert np.abs(g[a:b]).max() < 1e-12          # force balance + zero net torque by symmetry
    nominal = {0: [0.34, 0.19, -0.42], 1: [0.34, -0.19, -0.42], 2: [-0.34, 0.19, -0.42], 3: [-0.34, -0.19, -0.42]}
    lo, up = P.bounds()
    for e in range(4):
        a, b = sets["rangeofmotion-%d" % e]
        assert np.allclose(g[a:b].reshape(-1, 3), nominal[e], atol=1e-14)   # R = I: g = p - c
        assert np.allclose((lo[a:b] + up[a:b]).reshape(-1, 3) / 2, nominal[e], atol=1e-15)
        a, b = sets["terrain-ee-motion_%d" % e]
        assert np.array_equal(g[a:b], np.zeros(b - a))                       # p.z - 0
        for r in range(a, b):
            assert np.array_equal(va[rp[r]:rp[r + 1]], [-0.0, -0.0, 1.0])    # (-h_x, -h_y, 1)
        a, b = sets["force-ee-force_%d" % e]
        fz = 29.5 * 9.80665 / 4
        assert np.allclose(g[a:b].reshape(-1, 5), [fz, -0.5 * fz, 0.5 * fz, -0.5 * fz, 0.5 * fz], rtol=1e-15)
        blk = va[rp[a]:rp[a + 5]].reshape(5, 5)   # cols: foothold x,y | f x,y,z
        assert np.array_equal(blk[:, :2], np.zeros((5, 2)))
        assert np.array_equal(blk[:, 2:], [[0, 0, 1], [1, 0, -0.5], [1, 0, 0.5], [0, 1, -0.5], [0, 1, 0.5]])
    assert np.array_equal(lo[sets["dynamic"][0]:sets["dynamic"][1]], np.zeros(sets["dynamic"][1] - sets["dynamic"][0]))


def _set_rows(P):
    out, r0 = {}, 0
    for name, rows in P.con_sets:
        out[name] = (r0, r0 + rows)
        r0 += rows
    return out


def test_spline_acc_known_answers():
    """spline_acc_constraint.cc:49-88: base nodes sampled from ONE global cubic have continuous
    acceleration -> g = 0; the Jacobian rows are the kAcc weights of polynomial.cc:140-234 at t=T and t=0,
    subtracted as sparse rows (the shared node's position entry cancels but stays in the pattern)."""
    P = ob.OracleProblem("monoped", "flat", [[0.4, 0.2, 0.43]], [1], duration_base_poly=0.1, constraint_sets=ob.SETS_TOWR_DEFAULT)
    sets = _set_rows(P)
    durs = [0.1] * 10 + [0.03]               # parameters.cc:82-98: last polynomial takes the remainder
    assert sets["splineacc-base-lin"][1] - sets["splineacc-base-lin"][0] == 3 * (len(durs) - 1)
    t_nodes = np.concatenate([[0.0], np.cumsum(durs)])
    x = np.random.default_rng(5).normal(size=P.n)
    coef = np.random.default_rng(6).normal(size=(2, 3, 4))   # [lin|ang][dim][power]
    for which in range(2):
        for k, t in enumerate(t_nodes):
            for d in range(3):
                c = coef[which, d]
                x[which * 6 * len(t_nodes) + 6 * k + d] = c @ [1, t, t * t, t**3]
                x[which * 6 * len(t_nodes) + 6 * k + 3 + d] = c @ [0, 1, 2 * t, 3 * t * t]
    g, rp, ci, va = P.eval(x)
    for which, name in enumerate(("splineacc-base-lin", "splineacc-base-ang")):
        a, b = sets[name]
        assert np.abs(g[a:b]).max() < 1e-9            # accelerations ~ 1e1, coefficients ~ 6/T^2 = 6.7e3
        for j in range(len(durs) - 1):
            Tp, Tn = durs[j], durs[j + 1]
            expect = [6 / Tp**2, 2 / Tp, -6 / Tp**2 + 6 / Tn**2, 4 / Tp + 4 / Tn, -6 / Tn**2, 2 / Tn]
            for d in range(3):
                r = a + 3 * j + d
                cols = which * 6 * len(t_nodes) + 6 * j + d + 3 * np.arange(6)
                assert np.array_equal(ci[rp[r]:rp[r + 1]], cols)
                assert np.allclose(va[rp[r]:rp[r + 1]], expect, rtol=1e-12, atol=1e-9)
    lo, up = P.bounds()
    a, b = sets["splineacc-base-lin"][0], sets["splineacc-base-ang"][1]
    assert not lo[a:b].any() and not up[a:b].any()
    # a velocity kink at one node shows up in exactly the two junctions next to it... and at the node itself
    x2 = x.copy()
    x2[6 * 4 + 3 + 1] += 1.0                          # base-lin node 4, vy
    g2 = P.values(x2)
    a, b = sets["splineacc-base-lin"]
    changed = np.nonzero(np.abs(g2[a:b] - g[a:b]) > 1e-9)[0]
    assert changed.tolist() == [3 * 2 + 1, 3 * 3 + 1, 3 * 4 + 1]       # junctions 2,3,4 (nodes 3,4,5), dim y
    assert np.allclose((g2 - g)[a:b][changed], [2 / 0.1, 8 / 0.1, 2 / 0.1], rtol=1e-9)


def test_swing_known_answers():
    """swing_constraint.cc:58-121: swing node at the xy midpoint of its neighbours, xy velocity =
    distance / 0.3 s (swing_constraint.h:68)."""
    P = ob.OracleProblem("monoped", "flat", [[0.4, 0.2, 0.4, 0.3, 0.2]], [1], constraint_sets=ob.SETS_TOWR_DEFAULT)
    sets = _set_rows(P)
    a, b = sets["swing-ee-motion_0"]
    assert b - a == 2 * 4                         # two swing phases x one middle node x {x,y} x {pos,vel}
    off = dict((n, 0) for n, _ in P.var_sets)
    o = 0
    for n, sz in P.var_sets:
        off[n] = o
        o += sz
    m = off["ee-motion_0"]                        # stance(3) | swing node px vx py vy pz (5) | stance(3) | ...
    x = np.zeros(P.n)
    x[m:m + 3] = [0.0, 0.0, 0.0]
    x[m + 8:m + 11] = [0.3, 0.6, 0.0]
    x[m + 3:m + 8] = [0.15, 1.0, 0.3, 2.0, 0.07]  # midpoint, (0.3, 0.6) / 0.3 s
    x[m + 16:m + 19] = [0.5, 0.5, 0.0]
    x[m + 11:m + 16] = [0.45, 0.0, 0.5, 0.1, 0.1] # x pos off by +0.05, x vel by -(0.2/0.3), y pos by -0.05, y vel +0.1+1/3
    g, rp, ci, va = P.eval(x)
    assert np.abs(g[a:a + 4]).max() < 1e-15
    assert np.allclose(g[a + 4:b], [0.05, -0.2 / 0.3, -0.05, 0.1 + 0.1 / 0.3], atol=1e-15)
    for k in range(2):
        base = m + 8 * k
        rows = [va[rp[r]:rp[r + 1]] for r in range(a + 4 * k, a + 4 * k + 4)]
        cols = [ci[rp[r]:rp[r + 1]].tolist() for r in range(a + 4 * k, a + 4 * k + 4)]
        assert cols == [[base, base + 3, base + 8], [base, base + 4, base + 8], [base + 1, base + 5, base + 9], [base + 1, base + 6, base + 9]]
        assert np.array_equal(rows[0], [-0.5, 1.0, -0.5]) and np.array_equal(rows[2], [-0.5, 1.0, -0.5])
        assert np.array_equal(rows[1], [1 / 0.3, 1.0, -1 / 0.3]) and np.array_equal(rows[3], [1 / 0.3, 1.0, -1 / 0.3])
    lo, up = P.bounds()
    assert not lo[a:b].any() and not up[a:b].any()
    # a foot that starts in swing has no previous node: refused (the reference would throw in at())
    with pytest.raises(RuntimeError):
        ob.OracleProblem("monoped", "flat", [[0.3, 0.5, 0.3, 0.5]], [0], constraint_sets=ob.SETS_TOWR_DEFAULT)


def test_optimised_timings_known_answers():
    """Parameters::OptimizePhaseDurations (parameters.cc:76-80): ee-schedule<e> variable sets of n_phases-1
    durations bounded by [0.2, 1.0] (parameters.cc:52), every Jacobian row of an ee spline holds ALL variables
    of its set (phase_spline.cc:44-51), totalduration-<e> = sum of the optimised durations in
    [0.1, T - 0.2] (total_duration_constraint.cc:50-72)."""
    pd, con = ob.gait(2, 0, 2.0)
    F = ob.OracleProblem("biped", "flat", pd, con, constraint_sets=ob.SETS_TOWR_DEFAULT)
    P = ob.OracleProblem("biped", "flat", pd, con, constraint_sets=ob.SETS_TOWR_DEFAULT | ob.SET_TOTAL_TIME)
    assert P.var_sets[:-2] == F.var_sets and P.var_sets[-2:] == [("ee-schedule0", len(pd[0]) - 1), ("ee-schedule1", len(pd[1]) - 1)]
    assert P.con_sets[:-2] == F.con_sets and P.con_sets[-2:] == [("totalduration-0", 1), ("totalduration-1", 1)]
    ee = [[0.0, 0.2, 0.0], [0.0, -0.2, 0.0]]
    x = P.initial_guess([0, 0, 0.65], [0, 0, 0], [1, 0, 0.65], [0, 0, 0], ee)
    xf = F.initial_guess([0, 0, 0.65], [0, 0, 0], [1, 0, 0.65], [0, 0, 0], ee)
    assert np.array_equal(x[:F.n], xf) and np.array_equal(x[F.n:], np.concatenate([pd[0][:-1], pd[1][:-1]]))
    lo, up = P.variable_bounds(np.zeros(12), np.zeros(12), ee)
    assert np.all(lo[F.n:] == 0.2) and np.all(up[F.n:] == 1.0)
    g, rp, ci, va = P.eval(x)
    gf, rpf, cif, vaf = F.eval(xf)
    # same spline values at the unperturbed durations: identical g, and the last rows are the duration sums
    assert np.abs(g[:F.m] - gf).max() < 1e-12 * np.abs(gf).max()
    assert g[F.m] == pytest.approx(sum(pd[0][:-1]), abs=1e-15) and g[F.m + 1] == pytest.approx(sum(pd[1][:-1]), abs=1e-15)
    clo, cup = P.bounds()
    assert np.array_equal(clo[F.m:], [0.1, 0.1]) and np.allclose(cup[F.m:], [2.0 - 0.2] * 2, atol=1e-15)
    r = F.m
    assert ci[rp[r]:rp[r + 1]].tolist() == list(range(F.n, F.n + len(pd[0]) - 1)) and np.all(va[rp[r]:rp[r + 1]] == 1.0)
    # pattern: a rangeofmotion-0 row = 12 base-lin + 12 (8 for x) base-ang + all ee-motion_0 variables + all durations
    sets = _set_rows(P)
    sizes = dict(P.var_sets)
    a, _ = sets["rangeofmotion-0"]
    assert rp[a + 2] - rp[a + 1] == 12 + 12 + sizes["ee-motion_0"] + sizes["ee-schedule0"]
    assert rp[a + 1] - rp[a] == 12 + 8 + sizes["ee-motion_0"] + sizes["ee-schedule0"]
    # the non-zero values of the denser rows are the fixed-timing values, everything else is an explicit zero
    a, b = sets["dynamic"]
    af, _ = _set_rows(F)["dynamic"]
    for k in (0, 7, 13):
        for i in range(6):
            row = dict(zip(ci[rp[a + 6 * k + i]:rp[a + 6 * k + i + 1]].tolist(), va[rp[a + 6 * k + i]:rp[a + 6 * k + i + 1]]))
            rowf = dict(zip(cif[rpf[af + 6 * k + i]:rpf[af + 6 * k + i + 1]].tolist(), vaf[rpf[af + 6 * k + i]:rpf[af + 6 * k + i + 1]]))
            for c, v in rowf.items():
                assert row[c] == pytest.approx(v, rel=1e-12, abs=1e-12)
            assert all(v == 0.0 for c, v in row.items() if c < F.n and c not in rowf)


def test_base_motion_known_answers():
    """base_motion_constraint.cc:38-99: rows 6k+{AX,AY,AZ,LX,LY,LZ} = base-ang / base-lin position at
    t_k (dt = duration_base_polynomial_/4), roll/pitch within +-0.05 rad, z within [z_init-0.02, z_init+0.1]."""
    P = ob.OracleProblem("monoped", "flat", [[0.4, 0.2, 0.4]], [1], constraint_sets=ob.SET_BASE_ROM, base_z_init=0.58)
    assert P.con_sets == [("baseMotion", 6 * (int(np.floor(1.0 / 0.025)) + 2))]
    x = P.initial_guess([0.0, 0.1, 0.58], [0.01, 0.02, 0.3], [1.0, 0.3, 0.58], [0.03, 0.0, 0.5], [[0, 0, 0]])
    g, rp, ci, va = P.eval(x)
    lo, up = P.bounds()
    K = P.m // 6
    t = np.minimum(np.arange(K) * 0.025, 1.0)   # the linear initial guess: position = start + t/T * (goal - start)
    assert np.allclose(g.reshape(K, 6)[:, 3], t * 1.0, atol=1e-12) and np.allclose(g.reshape(K, 6)[:, 2], 0.3 + t * 0.2, atol=1e-12)
    assert np.allclose(g.reshape(K, 6)[:, 5], 0.58, atol=1e-12)   # z: terrain height + nominal stance height
    assert np.array_equal(lo.reshape(K, 6)[0], [-0.05, -0.05, -1e20, -1e20, -1e20, 0.58 - 0.02])
    assert np.array_equal(up.reshape(K, 6)[0], [0.05, 0.05, 1e20, 1e20, 1e20, 0.58 + 0.1])
    assert np.all(np.diff(rp) == 4)
    nb = dict(P.var_sets)["base-lin"]
    assert np.all(ci[rp[0]:rp[3]] >= nb) and np.all(ci[rp[3]:rp[6]] < nb)   # ang rows -> base-ang, lin rows -> base-lin
    assert np.allclose(va[rp[0]:rp[1]], [1, 0, 0, 0], atol=1e-15)            # t = 0: only p0 has weight


def test_time_grid_and_sizes_follow_the_reference_rules():
    """time_discretization_constraint.cc:37-50: K = floor(T/dt)+2 with a (near-)duplicate last node;
    sizes of the BASELINE configs as derived in SURVEY App. B."""
    P = ob.OracleProblem("monoped", "flat", [[0.4, 0.2, 0.4, 0.2, 0.4, 0.2, 0.2]], [1])
    assert (P.n, P.m, P.nnz) == (339, 273, 4672)
    assert dict(P.con_sets) == {"terrain-ee-motion_0": 10, "dynamic": 22 * 6, "rangeofmotion-0": 27 * 3,
                                "force-ee-force_0": 50}
    pd, con = ob.gait(4, 1, 2.0)
    assert [len(p) for p in pd] == [7, 9, 9, 7] and con == [1, 1, 1, 1]
    Q = ob.OracleProblem("anymal", "flat", pd, con, dt_dynamic=2.0 / 198.5, dt_rom=2.0 / 198.5)
    assert (Q.n, Q.m, Q.nnz) == (640, 3866, 102896)
    # towr's whole default list adds 2 x 3 x 19 splineacc rows (6 values each) and 4 rows (3 values) per swing node
    Qf = ob.OracleProblem("anymal", "flat", pd, con, dt_dynamic=2.0 / 198.5, dt_rom=2.0 / 198.5, constraint_sets=ob.SETS_TOWR_DEFAULT)
    n_swing = 3 + 4 + 4 + 3
    assert (Qf.n, Qf.m, Qf.nnz) == (640, 3866 + 114 + 4 * n_swing, 102896 + 684 + 12 * n_swing)
    assert [s for _, s in Q.var_sets] == [126, 126, 27, 35, 35, 27, 60, 72, 72, 60]
    pd, con = ob.gait(2, 0, 2.0)
    Bp = ob.OracleProblem("biped", "flat", pd, con, dt_dynamic=2.0 / 98.5, dt_rom=2.0 / 98.5)
    assert (Bp.n, Bp.m, Bp.nnz) == (466, 1346, 29676)


@pytest.mark.parametrize("robot,terrain,n_ee,combo,T", [("monoped", "flat", 1, 0, 2.0), ("biped", "block", 2, 1, 1.6),
                                                        ("anymal", "gap", 4, 3, 1.4), ("hyq", "chimney_lr", 4, 4, 1.5)])
def test_oracle_jacobian_vs_finite_differences(robot, terrain, n_ee, combo, T):
    pd, con = ob.gait(n_ee, combo, T)
    P = ob.OracleProblem(robot, terrain, pd, con)
    rng = np.random.default_rng(5)
    x = rng.normal(size=P.n) * 0.3
    g, rp, ci, va = P.eval(x)
    J = np.zeros((P.m, P.n))
    rows = np.repeat(np.arange(P.m), np.diff(rp))
    J[rows, ci] = va
    h = 1e-6
    Jfd = np.empty_like(J)
    for j in range(P.n):
        e = np.zeros(P.n)
        e[j] = h
        Jfd[:, j] = (P.values(x + e) - P.values(x - e)) / (2 * h)
    mask = np.zeros_like(J, dtype=bool)
    mask[rows, ci] = True
    assert np.abs(Jfd[~mask]).max() == 0.0          # nothing outside the pattern
    qm = np.zeros_like(J, dtype=bool)
    qk = quirk_mask(P, terrain, x, rp, ci)
    qm[rows[qk], ci[qk]] = True
    err = np.abs(J - Jfd) / np.maximum(1.0, np.abs(Jfd))
    assert err[~qm].max() < 2e-5


def _planar_grid_map(sx=40, sy=30, res=0.05, pos=(1.0, 0.2), a=0.3, b=-0.2, c=0.1):
    """elevation[i, j] of the plane h = a x + b y + c at grid_map's cell centres (x falls with i, y falls with j)."""
    i = np.arange(sx)[:, None]
    j = np.arange(sy)[None, :]
    cx = pos[0] + 0.5 * sx * res - (i + 0.5) * res
    cy = pos[1] + 0.5 * sy * res - (j + 0.5) * res
    return (a * cx + b * cy + c).astype(np.float32), res, pos


def test_grid_map_terrain_known_answers():
    """`Grid` (include/towr/terrain/grid_height_map.h:15-60) over grid_map's published atPosition(INTER_LINEAR):
    bilinear sampling reproduces a plane exactly (up to the float the reference stores the height in), cell centres
    return the cell, the half-cell border band falls back to the nearest cell, outside is FLT_MAX; slopes are central
    float differences over eps = resolution / 6."""
    el, res, pos = _planar_grid_map()
    P = ob.OracleProblem("anymal", "grid_map", *ob.gait(4, 1, 2.0), grid_map=(el, res, pos))
    f32 = np.float32
    # interior: the plane, rounded to float; slopes exact to float resolution of the heights / (2 eps)
    for x, y in [(1.0, 0.2), (0.3, 0.0), (1.93, 0.5), (0.52, -0.31), (1.2345, 0.4321)]:
        h, hx, hy = P.terrain_probe(x, y)
        want = 0.3 * x - 0.2 * y + 0.1
        assert abs(h - want) <= 1e-6 and h == float(f32(h))          # a float, widened
        assert abs(hx - 0.3) <= 2e-5 and abs(hy + 0.2) <= 2e-5
    # a cell centre returns that cell's value (weights 1,0,0,0)
    cx = pos[0] + 0.5 * 40 * res - (7 + 0.5) * res
    cy = pos[1] + 0.5 * 30 * res - (11 + 0.5) * res
    assert abs(P.terrain_probe(cx, cy)[0] - float(el[7, 11])) <= 1e-7
    # border band (less than half a cell from the map edge): nearest cell, hence zero slope (eps < half a cell)
    h, hx, hy = P.terrain_probe(0.01, -0.54)
    assert h == float(el[39, 29]) and hx == 0.0 and hy == 0.0
    h, hx, hy = P.terrain_probe(1.99, 0.94)
    assert h == float(el[0, 0]) and hx == 0.0 and hy == 0.0
    # outside: numeric_limits<float>::max(); FLT_MAX - FLT_MAX = 0 slope; at the edge one side is inside
    fmax = float(np.finfo(np.float32).max)
    assert tuple(P.terrain_probe(2.1, 0.0)) == (fmax, 0.0, 0.0)
    h, hx, hy = P.terrain_probe(0.0, 0.0)        # x = lower edge: position - length/2 hits the half-open bound
    assert h == fmax and hx < -1e39 and hy == 0.0
    # a 2x2 map: the bilinear weights at the centre are 1/4 each, evaluated in double, rounded to float
    el2 = np.array([[1.0, 2.0], [4.0, 8.0]], dtype=np.float32)
    P2 = ob.OracleProblem("monoped", "grid_map", [[0.4, 0.2, 0.4]], [1], grid_map=(el2, 1.0, (0.0, 0.0)))
    assert P2.terrain_probe(0.0, 0.0)[0] == 3.75
    # x = 0.25: cells i=0 (x centre +0.5) and i=1 (x centre -0.5): rx = 0.75 towards i=0
    assert P2.terrain_probe(0.25, 0.0)[0] == float(f32(0.5 * (0.75 * (1 + 2) + 0.25 * (4 + 8))))


def test_nearest_plane_known_answers():
    """fpowr PlanarRegionsToPolygons / NearestPlaneLookup (nearest_plane_lookup.h:20-85) with boost::geometry's
    distance(point, polygon) restated: 0 inside and on the boundary, distance to the nearest boundary segment outside,
    the first polygon wins a tie, -1 without polygons; the boundary points are walked as given (an unclosed ring has no
    closing edge: a point next to the missing edge is 'outside' and measured to the edges that exist)."""
    sq = np.array([[0, 0], [0, 1], [1, 1], [1, 0], [0, 0]], float)          # closed, clockwise
    xy = np.concatenate([sq, sq + [3, 0], (sq + [1.5, 2.0])[::-1]])         # third one counter-clockwise
    start = [0, 5, 10, 15]
    near = lambda x, y: ob.nearest_plane(xy, start, x, y)
    assert near(0.5, 0.5) == 0 and near(3.2, 0.9) == 1 and near(2.0, 2.5) == 2       # inside (either orientation)
    assert near(1.0, 0.5) == 0 and near(3.0, 0.0) == 1                                # on an edge / a corner
    assert near(1.9, 0.5) == 0 and near(2.1, 0.5) == 1 and near(2.0, 0.5) == 0        # outside; the tie goes to the first
    assert near(2.0, 1.4) == 2                                                        # 0.6 below polygon 2, ~1.08 from 0 and 1
    assert near(-7.0, -7.0) == 0 and near(40.0, 3.0) == 1
    assert ob.nearest_plane(np.zeros((0, 2)), [0], 0.3, 0.3) == -1
    # corner distance: point (1.3, 1.4) is sqrt(0.09 + 0.16) = 0.5 from corner (1, 1) of polygon 0, 0.6 below polygon 2
    assert near(1.3, 1.4) == 0 and near(1.6, 1.45) == 2
    # unclosed ring (last point not repeated): the edge (1,0)-(0,0) does not exist
    opn = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], float)
    far = sq + [0.0, -3.0]
    xy2, st2 = np.concatenate([opn, far]), [0, 4, 9]
    assert ob.nearest_plane(xy2, st2, 0.5, 0.9) == 0
    # (0.5, -1.4): 1.4 + from the existing edges of the open ring (corners (0,0) / (1,0): sqrt(0.25 + 1.96) = 1.487),
    # 0.6 above the closed square below -> the square; a closing edge would have made it 1.4 vs 0.6 as well
    assert ob.nearest_plane(xy2, st2, 0.5, -1.4) == 1
    # fewer than 4 points: never 'inside' (minimum ring size), distance to its segments
    tri = np.array([[0, 0], [2, 0], [1, 2]], float)
    assert ob.nearest_plane(np.concatenate([sq + [5, 5], tri]), [0, 5, 8], 1.0, 0.5) == 1
    # PlanarRegionsToPolygons: yaw 90 degrees maps local (1, 0) to world (0, 1); roll 180 degrees flips y; + position
    q_yaw = [0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)]
    w = ob.planes_world_xy([[1, 2, 3] + q_yaw, [0, 0, 0, 1, 0, 0, 0]], [[1, 0], [0, 1], [0.25, 0.5]], [0, 2, 3])
    assert np.allclose(w, [[1, 3], [0, 2], [0.25, -0.5]], rtol=0, atol=1e-15)
    # a non-unit quaternion is normalised by 2 / |q|^2 (tf Matrix3x3::setRotation)
    w2 = ob.planes_world_xy([[0, 0, 0, 0, 0, 3 * np.sin(0.3), 3 * np.cos(0.3)]], [[1, 0]], [0, 1])
    assert np.allclose(w2, [[np.cos(0.6), np.sin(0.6)]], rtol=0, atol=1e-15)


def test_third_party_restatements_against_independent_implementations():
    """tf's quaternion -> matrix, boost::geometry's point-polygon distance and grid_map's bilinear atPosition are absent
    from /root/reference (third party), so the oracle restates them from their published algorithms.  Besides the hand
    known-answers above they are cross-checked here against implementations that share no code with the oracle: scipy's
    Rotation, a brute-force numpy point-in-polygon / segment-distance, scipy's RegularGridInterpolator."""
    from scipy.interpolate import RegularGridInterpolator
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(2024)
    # (1) PlanarRegionsToPolygons: R(q) (x, y, 0) + position, any (non-unit) quaternion
    for _ in range(40):
        q = rng.normal(size=4) * rng.uniform(0.2, 3.0)
        pos = rng.uniform(-2, 2, size=3)
        pts = rng.uniform(-1, 1, size=(5, 2))
        got = ob.planes_world_xy([list(pos) + list(q)], pts, [0, 5])
        R = Rotation.from_quat(q).as_matrix()                       # scipy: [x, y, z, w], normalises
        want = (R @ np.c_[pts, np.zeros(5)].T).T[:, :2] + pos[:2]
        assert np.abs(got - want).max() <= 1e-13

    # (2) nearest polygon: 0 inside (even-odd ray casting), else the distance to the closest boundary segment
    def brute(polys, px, py):
        best, arg = np.inf, -1
        for i, P in enumerate(polys):
            inside = False
            for (x0, y0), (x1, y1) in zip(P[:-1], P[1:]):
                if (y0 > py) != (y1 > py) and px < x0 + (py - y0) * (x1 - x0) / (y1 - y0):
                    inside = not inside
            d = 0.0
            if not inside:
                d = np.inf
                for a, b in zip(P[:-1], P[1:]):
                    ab, ap = b - a, np.array([px, py]) - a
                    t = np.clip(ap @ ab / max(ab @ ab, 1e-300), 0.0, 1.0)
                    d = min(d, float(np.hypot(*(ap - t * ab))))
            if d < best:
                best, arg = d, i
        return arg, best

    for trial in range(30):
        polys = []
        for _ in range(int(rng.integers(2, 7))):
            k = int(rng.integers(3, 9))
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            rad = rng.uniform(0.2, 1.0, k)                           # star-shaped: concave corners happen
            P = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1) + rng.uniform(-3, 3, size=2)
            if trial % 2:
                P = P[::-1]                                          # either orientation
            polys.append(np.concatenate([P, P[:1]]))                 # closed ring
        xy = np.concatenate(polys)
        start = np.concatenate([[0], np.cumsum([len(P) for P in polys])])
        for _ in range(60):
            px, py = rng.uniform(-4.5, 4.5, size=2)
            arg, best = brute(polys, px, py)
            # skip near-ties and points within 1e-9 of a boundary (there the tie / boundary rules of boost decide)
            d_all = sorted(brute([P], px, py)[1] for P in polys)
            if d_all[1] - d_all[0] < 1e-9 or 0 < best < 1e-9:
                continue
            assert ob.nearest_plane(xy, start, px, py) == arg, (trial, px, py)

    # (3) Grid: bilinear interpolation between cell centres, strictly inside the map
    for _ in range(6):
        sx, sy = int(rng.integers(5, 30)), int(rng.integers(5, 30))
        res = float(rng.uniform(0.02, 0.2))
        pos = tuple(rng.uniform(-1, 1, size=2))
        el = rng.uniform(-0.5, 0.5, size=(sx, sy)).astype(np.float32)
        P = ob.OracleProblem("monoped", "grid_map", [[0.4, 0.2, 0.4]], [1], grid_map=(el, res, pos))
        cx = pos[0] + 0.5 * sx * res - (np.arange(sx) + 0.5) * res    # grid_map: cell (i, j) centre, descending in i, j
        cy = pos[1] + 0.5 * sy * res - (np.arange(sy) + 0.5) * res
        f = RegularGridInterpolator((cx[::-1], cy[::-1]), el[::-1, ::-1].astype(np.float64), method="linear")
        for _ in range(40):
            x = rng.uniform(cx[-1] + 1e-6, cx[0] - 1e-6)
            y = rng.uniform(cy[-1] + 1e-6, cy[0] - 1e-6)
            h = P.terrain_probe(x, y)[0]
            assert abs(h - float(f([[x, y]])[0])) <= 2e-7 and h == float(np.float32(h))


def test_initial_guess_samples_known_answer():
    """fpowr::ExtractInitialGuess (initial_guess_extractor.h:17-34).  At a base-spline node time the state is that
    node's variables: hopper, base polynomials of 0.1 s, variable layout [p0 v0 | p1 v1 | ...] per base set
    (nodes_variables_all.cc:45-61): t = 0.3 is node 3.  The controls hold the foot acceleration at 0..2, zeros at
    3..23 (one foot; twelve "joint torques"), the foot force at 24..26, zeros after; and they agree with GetTrajectory."""
    P = ob.OracleProblem("monoped", "flat", [[0.4, 0.2, 0.4, 0.2, 0.4, 0.2, 0.2]], [1])
    x = P.initial_guess([0, 0, 0.5], [0, 0, 0], [1, 0, 0.5], [0, 0, 0], [[0, 0, 0]])
    rng = np.random.default_rng(3)
    x = x + 0.05 * rng.standard_normal(x.size)
    rec = P.initial_guess_samples(x, [0.3, 0.0, 1.234])
    assert rec.shape == (3, 49) and list(rec[:, 0]) == [0.3, 0.0, 1.234]
    n_lin = 6 * 21    # 20 base polynomials of 0.1 s -> 21 nodes x (p, v) x 3 (parameters.cc:82-98)
    for row, node in ((0, 3), (1, 0)):
        lin, ang = x[6 * node:6 * node + 6], x[n_lin + 6 * node:n_lin + 6 * node + 6]
        assert np.allclose(rec[row, 1:4], lin[:3], rtol=0, atol=1e-13) and np.allclose(rec[row, 7:10], lin[3:], rtol=0, atol=1e-12)
        assert np.allclose(rec[row, 4:7], ang[:3], rtol=0, atol=1e-13) and np.allclose(rec[row, 10:13], ang[3:], rtol=0, atol=1e-12)
    assert np.array_equal(rec[:, 16:37], np.zeros((3, 21))) and np.array_equal(rec[:, 40:], np.zeros((3, 9)))
    traj = P.sample_trajectory(x, 0.1)     # [t | lin p v a | quaternion | omega | omega_dot | contact, ee p v a, force]
    assert abs(traj[3, 0] - 0.3) < 1e-12
    assert np.allclose(rec[0, 13:16], traj[3, 20 + 7:20 + 10], rtol=1e-12, atol=1e-12)
    assert np.allclose(rec[0, 37:40], traj[3, 20 + 10:20 + 13], rtol=1e-12, atol=1e-12)
    assert np.allclose(rec[0, 1:4], traj[3, 1:4], rtol=1e-12, atol=1e-12)


def test_contact_plan_known_answer():
    """fpowr::ExtractFootstepPlan (footstep_plan_extractor.h:69-133): footstep states where the contact flags change.
    Hopper phases {0.4,0.2,0.4,0.2,0.4,0.2,0.2} starting in contact, sampled every 0.01 s: states start at the first
    sample at or after every phase boundary (t accumulated, IsContactPhase uses the previous phase at a junction)."""
    P = ob.OracleProblem("monoped", "flat", [[0.4, 0.2, 0.4, 0.2, 0.4, 0.2, 0.2]], [1])
    x = P.initial_guess([0, 0, 0.5], [0, 0, 0], [1, 0, 0.5], [0, 0, 0], [[0, 0, 0]])
    plan = P.contact_plan(x, 0.01, 2.0)
    assert plan.shape == (7, 6)
    assert list(plan[:, 2]) == [1, 0, 1, 0, 1, 0, 1]
    bounds = np.cumsum([0.4, 0.2, 0.4, 0.2, 0.4, 0.2])
    # GetSegmentID: the previous phase still holds AT the boundary (eps 1e-10) -> the change shows one sample later
    assert plan[0, 0] == 0.0 and np.all(plan[1:, 0] > bounds - 1e-9) and np.all(plan[1:, 0] < bounds + 0.0101)
    assert np.allclose(plan[:-1, 1], np.diff(plan[:, 0])) and plan[-1, 1] == pytest.approx(2.0 - plan[-1, 0])
    assert np.allclose(plan[:, 3:], [[xx, 0, 0] for xx in plan[:, 3]])   # flat ground: the foot stays at z = 0
