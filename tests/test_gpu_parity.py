"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle and the golden
fixtures on identical inputs; size-independent properties at BASELINE's full batch size."""
import glob
import os

import numpy as np
import pytest

import towr_amd as ta
from tests.common import (Case, assert_parity, baseline_cases, hopper_schedule, k_params, parity_violations,
                          random_case, row_scale)

pytestmark = pytest.mark.gpu


def _eval_case(case, xs):
    batch = ta.Batch([case.S], [0] * len(xs), device=0)
    g, j = batch.eval_host(np.concatenate(xs))
    return batch, g, j


def _split(batch, g, j, p):
    return g[batch.g_off[p]:batch.g_off[p + 1]], j[batch.jac_off[p]:batch.jac_off[p + 1]]


@pytest.mark.parametrize("name", list(baseline_cases().keys()))
def test_baseline_configs_match_oracle(name):
    case = baseline_cases()[name]()
    goal = 2.0 if case.terrain != "flat" else 1.0
    xs = [case.x_guess(goal), case.x_perturbed(0, goal), case.x_perturbed(1, goal), case.x_wild(0)]
    batch, g, j = _eval_case(case, xs)
    for p, x in enumerate(xs):
        rg, _, _, rj = case.P.eval(x)
        gd, jd = _split(batch, g, j, p)
        assert_parity(case.S, gd, jd, rg, rj, "%s x[%d]" % (name, p))


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mp_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[3:-4] for p in GOLDEN])
def test_gpu_matches_mpmath_golden(path):
    """Committed fixtures of the independent 40-digit implementation (values; Jacobian entries that
    are true derivatives -- the terrain-basis quirk entries are pinned through the oracle)."""
    d = np.load(path)
    pd, o = [], 0
    for k in d["n_phases"]:
        pd.append(d["phase_durations"][o:o + k])
        o += k
    sets = int(d["constraint_sets"]) if "constraint_sets" in d.files else 27
    if sets & 64 and not getattr(ta, "SUPPORTS_OPTIMISED_TIMINGS", False):
        pytest.skip("optimised timings (SURVEY 8f #2): oracle and fixtures exist, the device path is next")
    dts = dict(dt_dynamic=float(d["dt_dynamic"]), dt_rom=float(d["dt_rom"])) if "dt_dynamic" in d.files else {}
    # (the fpowr fixture: Go1 on the `Grid` terrain, with its grid_map elevation layer)
    gm = dict(grid=(d["grid_elevation"], float(d["grid_resolution"]), tuple(d["grid_position"]))) if "grid_elevation" in d.files else {}
    case = Case(str(d["robot"]), str(d["terrain"]), ta.schedule(pd, list(d["contact_at_start"])), constraint_sets=sets,
                base_z_init=0.6, **dts, **gm)
    S = case.S
    batch, g, j = _eval_case(case, [d["x"]])
    assert np.abs(g - d["g"]).max() <= 1e-12 * np.abs(d["g"]).max()
    rows = np.repeat(np.arange(S.m), np.diff(S.row_ptr))
    ref = {(int(r), int(c)): v for r, c, v in zip(d["jac_row"], d["jac_col"], d["jac_val"])}
    refv = np.array([ref.get((r, c), 0.0) for r, c in zip(rows.tolist(), S.col_idx.tolist())])
    bad, err = parity_violations(j, refv, row_scale(S.row_ptr, refv))
    # allowed mismatches: force rows x foothold columns on the Gap (component-wise "derivative" quirk)
    force_rows = np.zeros(S.m, dtype=bool)
    for s in S.con_sets:
        if s["name"].startswith("force-"):
            force_rows[s["offset"]:s["offset"] + s["size"]] = True
    motion = [(v["offset"], v["offset"] + v["size"]) for v in S.var_sets if v["name"].startswith("ee-motion")]
    for k in bad:
        assert str(d["terrain"]) == "gap" and force_rows[rows[k]] and any(a <= S.col_idx[k] < b for a, b in motion), \
            "entry %d (row %d col %d): %.17g vs %.17g" % (k, rows[k], S.col_idx[k], j[k], refv[k])
    rg, _, _, rj = case.P.eval(d["x"])
    assert_parity(S, g, j, rg, rj, os.path.basename(path))


def test_large_ragged_batch_takes_the_persistent_node_kernel():
    """Batches of >= 2048 problems evaluate their terrain / force / splineacc / swing rows with the persistent
    node_chunk_kernel (per-family chunk lists, records two chunks ahead, x one chunk ahead) instead of one workgroup per
    problem: 2304 problems over three ragged structures (towr's default constraint list on Gap and Stairs, a K = 90 one with
    two splineacc chunks, the hot-path sets alone) against the oracle for a sample and against the SAME problems evaluated as
    a 768-problem batch (node_kernel path) for every value; NaN-prefilled outputs: every element written."""
    import torch
    specs = [("anymal", "gap", 1, 2.0, dict(constraint_sets=63)), ("hyq", "stairs", 0, 2.4, dict(constraint_sets=63, **k_params(2.4, 90))),
             ("go1", "slope", 3, 1.6, {})]
    cases = [Case(r, t, ta.gait_combo(4, c, T), **kw) for r, t, c, T, kw in specs]
    order = [p % 3 for p in range(2304)]
    xs_of = [[c.x_wild(i) for i in range(6)] + [c.x_perturbed(i, 2.0) for i in range(2)] for c in cases]
    xs = [xs_of[s][(p // 3) % 8] for p, s in enumerate(order)]
    outs = []
    for n in (2304, 768):
        batch = ta.Batch([c.S for c in cases], order[:n], device=0)
        x = torch.from_numpy(np.concatenate(xs[:n])).cuda()
        g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
        j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
        batch.profile_begin(1)   # per-kernel events: the separate launches (the fused launch has its own node role)
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        batch.profile_end()
        outs.append((batch, g.cpu().numpy(), j.cpu().numpy()))
    (b0, g0, j0), (b1, g1, j1) = outs
    assert np.isfinite(g0).all() and np.isfinite(j0).all()
    ng, nj = int(b1.g_off[-1]), int(b1.jac_off[-1])
    assert np.abs(g0[:ng] - g1).max() <= 1e-13 * np.abs(g1).max() and np.abs(j0[:nj] - j1).max() <= 1e-13 * np.abs(j1).max()
    for p in (0, 1, 2, 770, 1535, 2301, 2302, 2303):
        rg, _, _, rj = cases[order[p]].P.eval(xs[p])
        gd, jd = _split(b0, g0, j0, p)
        assert_parity(cases[order[p]].S, gd, jd, rg, rj, "problem %d" % p, x=xs[p])


def test_ragged_batch_of_distinct_structures():
    """Different gaits, horizons, terrains and robots in one launch (odd and even nnz offsets)."""
    specs = [("anymal", "gap", 0, 2.0, {}), ("anymal", "stairs", 1, 1.4, {}), ("hyq", "slope", 2, 2.4, {}),
             ("go1", "chimney", 3, 1.9, {}), ("anymal", "block", 4, 2.8, k_params(2.8, 64)),
             ("hyq", "chimney_lr", 1, 2.2, dict(polys_per_swing=3, polys_per_stance_force=2)),
             ("go1", "flat", 0, 1.3, dict(polys_per_swing=1, polys_per_stance_force=1, duration_base_poly=0.07))]
    cases = [Case(r, t, ta.gait_combo(4, c, T), **kw) for r, t, c, T, kw in specs]
    order = [0, 1, 2, 3, 4, 5, 6, 3, 1, 0, 6, 5]
    assert any(c.S.nnz % 2 for c in cases)
    batch = ta.Batch([c.S for c in cases], order, device=0)
    xs = [cases[s].x_wild(i) for i, s in enumerate(order)]
    g, j = batch.eval_host(np.concatenate(xs))
    for p, s in enumerate(order):
        rg, _, _, rj = cases[s].P.eval(xs[p])
        gd, jd = _split(batch, g, j, p)
        assert_parity(cases[s].S, gd, jd, rg, rj, "problem %d (structure %d)" % (p, s))


def test_fused_launch_equals_separate_launches():
    """Small batches go through one fused launch (eval_fused_kernel: rom, dyn and node roles in one grid), large or
    profiled ones through dyn_kernel / rom_kernel / node_kernel: same code per slice, so identical bits, and both equal
    to the oracle.  Ragged structures, odd and even offsets, NaN-prefilled outputs (every value written by both)."""
    import torch
    specs = [("anymal", "gap", 0, 2.0, {}), ("go1", "stairs", 3, 1.9, dict(constraint_sets=63)),
             ("hyq", "slope", 2, 2.4, k_params(2.4, 130)), ("anymal", "block", 4, 2.8, dict(constraint_sets=255, base_z_init=0.5))]
    cases = [Case(r, t, ta.gait_combo(4, c, T), **kw) for r, t, c, T, kw in specs]
    order = [0, 1, 2, 3, 2, 1, 0, 3, 3]
    batch = ta.Batch([c.S for c in cases], order, device=0)
    xs = [cases[s].x_wild(i) for i, s in enumerate(order)]
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    outs = []
    for profiled in (False, True):
        g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
        j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
        if profiled:
            batch.profile_begin(1)     # per-kernel events: the three separate launches
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if profiled:
            ms, n_evals = batch.profile_end()
            assert n_evals == 1 and all(v >= 0 for v in ms.values())
        outs.append((g.cpu().numpy(), j.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1]).all()
    for p, s in enumerate(order):
        rg, _, _, rj = cases[s].P.eval(xs[p])
        assert_parity(cases[s].S, *_split(batch, outs[0][0], outs[0][1], p), rg, rj, "problem %d" % p)


@pytest.mark.parametrize("robot,n_ee,combo", [("monoped", 1, 2), ("monoped", 1, 4), ("biped", 2, 1), ("biped", 2, 4)])
def test_other_leg_counts_and_gaits_starting_or_ending_in_flight(robot, n_ee, combo):
    case = Case(robot, "slope", ta.gait_combo(n_ee, combo, 2.1))
    xs = [case.x_wild(3), case.x_perturbed(2, 2.0)]
    batch, g, j = _eval_case(case, xs)
    for p, x in enumerate(xs):
        rg, _, _, rj = case.P.eval(x)
        assert_parity(case.S, *_split(batch, g, j, p), rg, rj, "%s c%d x[%d]" % (robot, combo, p))


@pytest.mark.parametrize("spec", [
    ("monoped", "flat", None, 2.0, dict(constraint_sets=63)),
    ("biped", "block", 0, 2.0, dict(constraint_sets=63)),
    ("anymal", "flat", 1, 2.0, dict(constraint_sets=63, **k_params(2.0, 200))),   # BASELINE C3 + the linear sets
    ("hyq", "gap", 2, 2.1, dict(constraint_sets=63, polys_per_swing=3, duration_base_poly=0.13)),
    ("go1", "slope", 0, 2.3, dict(constraint_sets=63, polys_per_swing=1)),
    ("anymal", "stairs", 0, 2.4, dict(constraint_sets=2 | 32)),
    ("biped", "flat", 1, 1.8, dict(constraint_sets=4)),
    ("anymal", "chimney", 1, 2.0, dict(constraint_sets=1 | 16 | 32)),
    ("monoped", "flat", None, 2.0, dict(constraint_sets=128, base_z_init=0.58)),                       # baseMotion alone
    ("anymal", "stairs", 1, 2.0, dict(constraint_sets=255, base_z_init=0.42, dt_base_motion=0.031)),   # every set
    ("biped", "gap", 0, 2.0, dict(constraint_sets=63 | 128, base_z_init=0.65)),
], ids=lambda s: "%s-%s-s%d" % (s[0], s[1], s[4]["constraint_sets"]))
def test_whole_default_constraint_list_and_subsets(spec):
    """towr's default constraints_ (parameters.cc:55-60) = hot path + splineacc-base-{lin,ang} + swing-*;
    any subset keeps the reference order.  Two problems per batch so that odd value offsets occur."""
    robot, terrain, combo, T, kw = spec
    n_ee = ta.model_preset(robot, terrain).n_ee
    sched = hopper_schedule() if combo is None else ta.gait_combo(n_ee, combo, T)
    case = Case(robot, terrain, sched, **kw)
    xs = [case.x_wild(4), case.x_perturbed(5, 2.0), case.x_guess(1.5)]
    batch, g, j = _eval_case(case, xs)
    for p, x in enumerate(xs):
        rg, _, _, rj = case.P.eval(x)
        assert_parity(case.S, *_split(batch, g, j, p), rg, rj, "%s x[%d]" % (spec[:3], p), x=x)


@pytest.mark.parametrize("spec", [
    ("monoped", "flat", None, 2.0, dict(constraint_sets=127)),
    ("biped", "stairs", 0, 2.0, dict(constraint_sets=127)),
    ("anymal", "gap", 1, 2.0, dict(constraint_sets=127)),
    ("anymal", "flat", 1, 2.0, dict(constraint_sets=127, **k_params(2.0, 200))),   # BASELINE C3 with optimised timings
    ("hyq", "slope", 3, 2.2, dict(constraint_sets=127, polys_per_swing=3, polys_per_stance_force=2)),
    ("go1", "block", 2, 1.9, dict(constraint_sets=2 | 64)),
    ("biped", "flat", 1, 1.8, dict(constraint_sets=8 | 64)),
], ids=lambda s: "%s-%s-s%d" % (s[0], s[1], s[4]["constraint_sets"]))
def test_optimised_phase_durations(spec):
    """Parameters::OptimizePhaseDurations: ee-schedule variables, x-dependent active polynomials, rows
    that hold all variables of every ee set (explicit zeros), duration columns, totalduration rows."""
    robot, terrain, combo, T, kw = spec
    n_ee = ta.model_preset(robot, terrain).n_ee
    sched = hopper_schedule() if combo is None else ta.gait_combo(n_ee, combo, T)
    case = Case(robot, terrain, sched, **kw)
    xs = [case.x_wild(6), case.x_perturbed(7, 2.0), case.x_guess(1.5), case.x_wild(8)]
    batch, g, j = _eval_case(case, xs)
    for p, x in enumerate(xs):
        rg, _, _, rj = case.P.eval(x)
        assert_parity(case.S, *_split(batch, g, j, p), rg, rj, "%s x[%d]" % (spec[:3], p), x=x)


def test_mixed_batch_of_fixed_and_optimised_timings():
    """Problems with and without optimised timings in one launch; stale buffer contents must not leak
    into the explicit zeros (the device path zero-fills before storing the non-zeros)."""
    import torch

    a = Case("anymal", "gap", ta.gait_combo(4, 1, 2.0), constraint_sets=127)
    b = Case("anymal", "gap", ta.gait_combo(4, 1, 2.0), constraint_sets=63)
    c = Case("anymal", "stairs", ta.gait_combo(4, 0, 2.4), constraint_sets=27 | 64)
    cases, order = [a, b, c], [0, 1, 2, 1, 0, 0, 2]
    batch = ta.Batch([k.S for k in cases], order, device=0)
    xs = [cases[s].x_wild(20 + i) for i, s in enumerate(order)]
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):  # second pass runs over the first pass's values
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    gh, jh = g.cpu().numpy(), j.cpu().numpy()
    assert np.isfinite(gh).all() and np.isfinite(jh).all()
    for p, s in enumerate(order):
        rg, _, _, rj = cases[s].P.eval(xs[p])
        assert_parity(cases[s].S, *_split(batch, gh, jh, p), rg, rj, "problem %d" % p, x=xs[p])


def _many_phases(n_ee, n_phases, seed):
    rng = np.random.default_rng(seed)
    durs = []
    for _ in range(n_ee):
        d = rng.uniform(0.12, 0.35, size=n_phases)
        durs.append(d * (3.1 / d.sum()))
    return ta.schedule(durs, [1] * n_ee)


@pytest.mark.parametrize("name,make", [
    ("standing_one_phase", lambda: Case("anymal", "flat", ta.schedule([[1.3]] * 4, [1] * 4))),
    ("two_nodes_dt_gt_T", lambda: Case("biped", "slope", ta.gait_combo(2, 0, 1.1), dt_dynamic=5.0, dt_rom=7.0)),
    ("max_phases_32", lambda: Case("anymal", "stairs", _many_phases(4, 31, 1), constraint_sets=63)),
    ("max_phases_32_timings", lambda: Case("biped", "gap", _many_phases(2, 31, 2), constraint_sets=127)),
    # largest expanded rows of the ABI: 4224 values per dynamic time node, a four-node pass takes 135 KB of LDS
    ("max_phases_32_timings_quad", lambda: Case("anymal", "stairs", _many_phases(4, 31, 3), constraint_sets=127)),
    ("long_horizon_K1000", lambda: Case("hyq", "block", ta.gait_combo(4, 1, 6.0), **k_params(6.0, 1000))),
    ("many_polys", lambda: Case("go1", "chimney_lr", ta.gait_combo(4, 2, 2.0), polys_per_swing=4, polys_per_stance_force=5)),
    ("fine_base_spline", lambda: Case("monoped", "gap", hopper_schedule(), duration_base_poly=0.013, constraint_sets=63)),
    ("two_phase_timings", lambda: Case("monoped", "flat", ta.schedule([[0.6, 0.5]], [1]), constraint_sets=27 | 64)),
], ids=lambda v: v if isinstance(v, str) else "")
def test_edge_sizes(name, make):
    """Smallest and largest structures: a single phase, two time nodes, the maximum phase count of the ABI
    (TWR_MAX_PHASES), a 1000-node horizon (many slices per set), many polynomials per phase."""
    case = make()
    xs = [case.x_wild(9), case.x_perturbed(10, 1.5)]
    batch, g, j = _eval_case(case, xs)
    for p, x in enumerate(xs):
        rg, _, _, rj = case.P.eval(x)
        assert_parity(case.S, *_split(batch, g, j, p), rg, rj, "%s x[%d]" % (name, p), x=x)
    # the values-only path (one lane per time node for "dynamic" / "rangeofmotion-*": items are cut at 64 time nodes or eight
    # polynomials of one spline; structures with more than 2046 variables or optimised timings keep the Jacobian kernels' cut)
    px, pg, pj = batch.host_buffers()
    px[:] = np.concatenate(xs)
    pg[:] = np.nan
    batch.eval_host_pinned(ta.EVAL_VALUES)
    assert np.abs(pg - g).max() <= 1e-12 * max(1.0, np.abs(g).max()), name


def test_optimised_timings_at_scale():
    """512 problems with optimised phase durations (1.6 GB of Jacobian values): every value is written
    (zeros included), copies of one x agree bit for bit wherever they sit, values-only calls leave the
    Jacobian buffer alone, sampled problems match the oracle."""
    import torch

    case = Case("anymal", "flat", ta.gait_combo(4, 1, 2.0), constraint_sets=127, **k_params(2.0, 200))
    S = case.S
    B, nb = 512, 16
    batch = ta.Batch([S], [0] * B, device=0)
    base = np.stack([case.x_perturbed(40 + i) for i in range(nb)])
    x = torch.from_numpy(np.tile(base, (B // nb, 1)).reshape(-1)).cuda()
    g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all()) and bool(torch.isnan(j).all())
    for _ in range(3):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(j).all())
    G, J = g.view(B, S.m), j.view(B, S.nnz)
    assert bool((G.view(B // nb, nb, S.m) == G[:nb]).all()) and bool((J.view(B // nb, nb, S.nnz) == J[:nb]).all())
    for p in (0, 7, 15):
        rg, _, _, rj = case.P.eval(base[p])
        assert_parity(S, G[p + nb * 3].cpu().numpy(), J[p + nb * 20].cpu().numpy(), rg, rj, "problem %d" % p, x=base[p])
    assert float((J[0] == 0).double().mean()) > 0.6   # most of the all-variables rows are explicit zeros


def test_page_locked_host_buffers():
    """twr_batch_host_buffers + twr_batch_eval_host: same results as the pageable path."""
    case = baseline_cases()["C2_biped_K100"]()
    batch = ta.Batch([case.S], [0, 0, 0], device=0)
    xs = np.concatenate([case.x_wild(i) for i in range(3)])
    g, j = batch.eval_host(xs)
    px, pg, pj = batch.host_buffers()
    assert px.size == xs.size and pg.size == g.size and pj.size == j.size
    px[:] = xs
    pg[:] = np.nan
    pj[:] = np.nan
    batch.eval_host_pinned()
    assert np.array_equal(pg, g) and np.array_equal(pj, j)
    px2, _, _ = batch.host_buffers()
    assert px2.ctypes.data == px.ctypes.data   # allocated once, owned by the batch


def test_page_locked_host_buffers_optimised_timings_and_flags():
    """The zero-copy branch of twr_batch_eval_host (kernels store g / jac straight into the batch's page-locked
    buffers and gather x from them) for the kernels test_page_locked_host_buffers does not reach: a ragged batch that
    mixes fixed and optimised timings with every constraint set, and the VALUES-only / JACOBIAN-only flag variants
    (untouched outputs must stay untouched); x copied from pageable memory while g / jac are zero-copy."""
    cases = [Case("anymal", "gap", ta.gait_combo(4, 1, 2.0), constraint_sets=127),
             Case("anymal", "stairs", ta.gait_combo(4, 2, 1.7), constraint_sets=63),
             Case("anymal", "flat", ta.gait_combo(4, 0, 2.2), constraint_sets=255, base_z_init=0.5)]
    order = [0, 1, 2, 0, 1]
    batch = ta.Batch([c.S for c in cases], order, device=0)
    xs = np.concatenate([cases[s].x_wild(70 + p) for p, s in enumerate(order)])
    g, j = batch.eval_host(xs)                      # pageable buffers: device outputs + copies
    for p, s in enumerate(order):
        rg, _, _, rj = cases[s].P.eval(xs[batch.x_off[p]:batch.x_off[p + 1]])
        assert_parity(cases[s].S, *_split(batch, g, j, p), rg, rj, "problem %d" % p, x=xs[batch.x_off[p]:batch.x_off[p + 1]])
    px, pg, pj = batch.host_buffers()
    px[:] = xs
    for flags, want_g, want_j in ((ta.EVAL_BOTH, True, True), (ta.EVAL_VALUES, True, False), (ta.EVAL_JACOBIAN, False, True)):
        pg[:] = np.nan
        pj[:] = np.nan
        batch.eval_host_pinned(flags)
        # (the single-output variants are other instruction sequences -- template instantiations / branches with their
        # own FMA contraction -- so they agree with the combined call to rounding, not bit for bit)
        same = np.array_equal if flags == ta.EVAL_BOTH else (lambda a, b: np.abs(a - b).max() <= 1e-13 * np.abs(b).max())
        assert same(pg, g) if want_g else np.isnan(pg).all(), flags
        assert same(pj, j) if want_j else np.isnan(pj).all(), flags
    # x from pageable memory, outputs zero-copy
    pg[:] = np.nan
    pj[:] = np.nan
    ta._check(ta.lib().twr_batch_eval_host(batch._h, ta._d(xs), ta._d(pg), ta._d(pj), ta.EVAL_BOTH))
    assert np.array_equal(pg, g) and np.array_equal(pj, j)


def test_random_structures_in_one_ragged_batch():
    """Seeded fuzz: 24 random problems (robots with the same leg count share a launch), every constraint-set
    mask, fixed and optimised timings mixed."""
    cases = [random_case(1000 + s) for s in range(60)]
    for n_ee in (1, 2, 4):
        group = [c for c in cases if c.S.n_ee == n_ee][:8]
        if not group:
            continue
        batch = ta.Batch([c.S for c in group], list(range(len(group))), device=0)
        xs = [c.x_wild(50 + i) for i, c in enumerate(group)]
        g, j = batch.eval_host(np.concatenate(xs))
        for p, c in enumerate(group):
            rg, _, _, rj = c.P.eval(xs[p])
            assert_parity(c.S, *_split(batch, g, j, p), rg, rj, "n_ee %d problem %d" % (n_ee, p), x=xs[p])
        # values only on the same ragged batch (fixed and optimised timings mixed: one problem that cannot take the
        # lane-per-node kernels keeps the whole batch on the Jacobian kernels' cut) and on its fixed-timings problems alone
        fixed = [i for i, c in enumerate(group) if not (c.params.constraint_sets & 64)]
        for sel in (list(range(len(group))), fixed):
            if not sel:
                continue
            sub = ta.Batch([group[i].S for i in sel], list(range(len(sel))), device=0)
            px, pg, pj = sub.host_buffers()
            px[:] = np.concatenate([xs[i] for i in sel])
            pg[:] = np.nan
            sub.eval_host_pinned(ta.EVAL_VALUES)
            want = np.concatenate([g[batch.g_off[i]:batch.g_off[i + 1]] for i in sel])
            assert np.abs(pg - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (n_ee, sel)


def test_gridded_terrain():
    """HeightMapFromCSV on the device: heights and edge slopes from a grid in device memory; two structures share
    one grid (uploaded once), a third has its own; footholds are placed on cell edges on purpose."""
    rng = np.random.default_rng(7)
    g1 = np.round(rng.uniform(0.0, 0.3, size=(18, 26)), 2)
    g2 = np.round(rng.uniform(0.0, 0.2, size=(5, 7)), 2)
    G1 = ta.TerrainGrid(g1)
    a = Case("anymal", "csv", ta.gait_combo(4, 1, 2.0), grid=G1, constraint_sets=63)
    b = Case("anymal", "csv", ta.gait_combo(4, 0, 2.4), grid=G1)
    c = Case("anymal", "csv", ta.gait_combo(4, 2, 1.8), grid=g2, constraint_sets=127)
    cases, order = [a, b, c], [0, 1, 2, 0, 2, 1]
    batch = ta.Batch([k.S for k in cases], order, device=0)
    xs = []
    for i, s in enumerate(order):
        x = cases[s].x_wild(70 + i)
        for vs in cases[s].S.var_sets:      # snap a third of the foothold coordinates onto / next to cell edges
            if vs["name"].startswith("ee-motion"):
                seg = x[vs["offset"]:vs["offset"] + vs["size"]]
                pick = rng.random(seg.size) < 0.35
                seg[pick] = np.round(seg[pick] / 0.17) * 0.17 + rng.uniform(-0.004, 0.004, size=pick.sum())
        xs.append(x)
    g, j = batch.eval_host(np.concatenate(xs))
    hit = 0
    for p, s in enumerate(order):
        rg, rp, ci, rj = cases[s].P.eval(xs[p])
        assert_parity(cases[s].S, *_split(batch, g, j, p), rg, rj, "problem %d" % p, x=xs[p])
        for cs in cases[s].S.con_sets:
            if cs["name"].startswith("terrain-"):
                blk = rj[rp[cs["offset"]]:rp[cs["offset"] + cs["size"]]].reshape(-1, 3)
                hit += int((np.abs(blk[:, :2]) > 1.0).sum())
    assert hit >= 5, "the inputs should exercise the edge slopes"


def test_trajectory_sampling():
    """twr_batch_sample vs the oracle's restatement of fpowr::GetTrajectory (footstep_plan_extractor.h:19-53):
    base state, quaternion, angular velocity / acceleration, contact flags, foot motion and forces every 10 ms,
    for fixed and optimised timings, with large Euler angles (both quaternion branches)."""
    import torch

    a = Case("anymal", "stairs", ta.gait_combo(4, 1, 2.0))
    b = Case("anymal", "gap", ta.gait_combo(4, 0, 2.4), constraint_sets=127)
    c = Case("anymal", "flat", ta.gait_combo(4, 3, 1.7), polys_per_swing=3, duration_base_poly=0.13)
    cases, order = [a, b, c], [0, 1, 2, 1, 0]
    batch = ta.Batch([k.S for k in cases], order, device=0)
    xs = [cases[s].x_wild(90 + i) for i, s in enumerate(order)]
    for i, s in enumerate(order):   # large rotations: trace(R) <= 0 for some samples
        for vs in cases[s].S.var_sets:
            if vs["name"] == "base-ang":
                xs[i][vs["offset"]:vs["offset"] + vs["size"]] *= 4.0
    dt = 0.01
    rec = 20 + 13 * 4
    counts = [cases[s].S.sample_count(dt) for s in order]
    stride = max(counts) * rec
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    out = torch.full((len(order) * stride,), float("nan"), dtype=torch.float64, device="cuda")
    batch.sample_device(x.data_ptr(), dt, out.data_ptr(), stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oh = out.cpu().numpy()
    neg_trace = 0
    for p, s in enumerate(order):
        ref = cases[s].P.sample_trajectory(xs[p], dt)
        assert ref.shape == (counts[p], rec)
        got = oh[p * stride:p * stride + counts[p] * rec].reshape(counts[p], rec)
        scale = np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1.0)
        assert np.array_equal(got[:, 0], ref[:, 0])                       # the accumulated sample times
        for e in range(4):
            assert np.array_equal(got[:, 20 + 13 * e], ref[:, 20 + 13 * e])   # contact flags
        err = np.abs(got - ref) / scale
        assert err.max() < 1e-10, "problem %d: field %d off by %.3e" % (p, int(err.max(axis=0).argmax()), err.max())
        neg_trace += int((4 * ref[:, 10] ** 2 - 1 <= 0).sum())          # trace(R) = 4 w^2 - 1
    assert neg_trace > 0


def test_initial_guess_samples():
    """twr_batch_initial_guess vs the oracle's restatement of fpowr::ExtractInitialGuess (initial_guess_extractor.h:
    17-34): base position / Euler angles / their rates, foot accelerations, twelve zeros, foot forces at arbitrary
    sample times (unsorted, incl. 0, T and polynomial junctions), for 4-, 2- and 1-legged robots, fixed and optimised
    timings; more than 64 times (two workgroups per problem)."""
    import torch

    for robot, n_ee, kw in (("anymal", 4, {}), ("anymal", 4, dict(constraint_sets=127)), ("biped", 2, {}), ("monoped", 1, {})):
        T = 2.0
        cases = [Case(robot, "stairs" if n_ee == 4 else "flat", ta.gait_combo(n_ee, 1 if n_ee > 1 else 2, T), **kw),
                 Case(robot, "flat", ta.gait_combo(n_ee, 0, T), duration_base_poly=0.13, **kw)]
        order = [0, 1, 0]
        batch = ta.Batch([k.S for k in cases], order, device=0)
        xs = [cases[s].x_wild(40 + i) for i, s in enumerate(order)]
        rng = np.random.default_rng(5)
        times = np.concatenate([[0.0, T, 0.1, 0.2, 0.13, 1.0], rng.uniform(0.0, T, 70)])
        stride = 49 * len(times) + 3
        x = torch.from_numpy(np.concatenate(xs)).cuda()
        tt = torch.from_numpy(times).cuda()
        out = torch.full((len(order) * stride,), float("nan"), dtype=torch.float64, device="cuda")
        batch.initial_guess_device(x.data_ptr(), tt.data_ptr(), len(times), out.data_ptr(), stride,
                                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        oh = out.cpu().numpy()
        for p, s in enumerate(order):
            ref = cases[s].P.initial_guess_samples(xs[p], times)
            got = oh[p * stride:p * stride + 49 * len(times)].reshape(len(times), 49)
            assert np.array_equal(got[:, 0], times)
            assert np.array_equal(got[:, 25:37], np.zeros((len(times), 12)))          # the "joint torques"
            assert np.array_equal(got[:, 13 + 3 * n_ee:25], np.zeros((len(times), 12 - 3 * n_ee)))   # absent feet
            assert np.isnan(oh[p * stride + 49 * len(times):(p + 1) * stride]).all()   # nothing written past the records
            scale = np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1.0)
            err = np.abs(got - ref) / scale
            assert err.max() < 1e-10, "%s problem %d: field %d off by %.3e" % (robot, p, int(err.max(axis=0).argmax()), err.max())


def test_foot_starting_in_swing():
    """ee_in_contact_at_start = false: first polynomial of ee-motion is a swing one, force starts at zero."""
    sched = ta.schedule([[0.3, 0.5, 0.3, 0.4], [0.6, 0.3, 0.6]], [0, 1])
    case = Case("biped", "gap", sched)
    xs = [case.x_wild(1)]
    batch, g, j = _eval_case(case, xs)
    rg, _, _, rj = case.P.eval(xs[0])
    assert_parity(case.S, g, j, rg, rj, "swing start")


def test_flags_select_outputs():
    import torch

    case = baseline_cases()["C1_hopper"]()
    batch = ta.Batch([case.S], [0, 0], device=0)
    x = torch.tensor(np.concatenate([case.x_wild(0), case.x_wild(1)]), device="cuda")
    g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_VALUES, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all()) and bool(torch.isnan(j).all())
    g.fill_(float("nan"))
    batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_JACOBIAN, st)
    torch.cuda.synchronize()
    assert bool(torch.isnan(g).all()) and bool(torch.isfinite(j).all())
    with pytest.raises(ta.TowrError):
        batch.eval_device(x.data_ptr(), 0, j.data_ptr(), ta.EVAL_BOTH, st)
    with pytest.raises(ta.TowrError):
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), 0, st)


def test_full_size_batch_properties():
    """BASELINE C3 at bench size (8192 problems): every output element is written exactly where the
    CSR layout says, duplicates of one x give bit-identical results anywhere in the batch, sampled
    problems match the oracle, and J is the derivative of the GPU's own g (directional FD)."""
    import torch

    case = baseline_cases()["C3_anymal_trot_K200"]()
    S = case.S
    B = 8192
    batch = ta.Batch([S], [0] * B, device=0)
    base = np.stack([case.x_perturbed(i) for i in range(32)])
    xh = np.tile(base, (B // 32, 1))
    x = torch.from_numpy(xh.reshape(-1)).cuda()
    g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(j).all())      # no element left unwritten
    G, J = g.view(B, S.m), j.view(B, S.nnz)
    assert bool((G.view(B // 32, 32, S.m) == G[:32]).all()) and bool((J.view(B // 32, 32, S.nnz) == J[:32]).all())
    for p in (0, 17, 31):
        rg, _, _, rj = case.P.eval(base[p])
        assert_parity(S, G[p + 32 * 5].cpu().numpy(), J[p + 32 * 100].cpu().numpy(), rg, rj, "problem %d" % p)
    # directional derivative: J d ~ (g(x + h d) - g(x - h d)) / 2h for the whole batch at once
    rng = np.random.default_rng(0)
    d = rng.normal(size=S.n)
    h = 1e-6
    gp = torch.empty_like(g)
    gm = torch.empty_like(g)
    xp = torch.from_numpy((xh + h * d).reshape(-1)).cuda()
    xm = torch.from_numpy((xh - h * d).reshape(-1)).cuda()
    batch.eval_device(xp.data_ptr(), gp.data_ptr(), 0, ta.EVAL_VALUES, st)
    batch.eval_device(xm.data_ptr(), gm.data_ptr(), 0, ta.EVAL_VALUES, st)
    torch.cuda.synchronize()
    fd = ((gp - gm) / (2 * h)).view(B, S.m)[:32].cpu().numpy()
    rows = np.repeat(np.arange(S.m), np.diff(S.row_ptr))
    Jd = np.zeros((32, S.m))
    Jn = J[:32].cpu().numpy()
    for p in range(32):
        np.add.at(Jd[p], rows, Jn[p] * d[S.col_idx])
    scale = np.maximum(np.abs(fd).max(axis=0, keepdims=True), 1.0)
    assert (np.abs(Jd - fd) / scale).max() < 1e-5


# ---------------------------------------------------------------- BASELINE C4 / C5: the enumerated sweeps
def _sweep_inputs(structs, model, first_seed=0):
    from bench import perturbed_inputs

    return [perturbed_inputs(S, model, 1, first_seed=first_seed + i)[0] for i, S in enumerate(structs)]


def _oracle_for(robot, terrain, S):
    from oracle import binding as ob

    p = S.params
    return ob.OracleProblem(robot, terrain, S.schedule.durations(), S.schedule.contact(), dt_dynamic=p.dt_dynamic,
                            dt_rom=p.dt_rom, duration_base_poly=p.duration_base_poly, polys_per_swing=p.polys_per_swing,
                            polys_per_stance_force=p.polys_per_stance_force, constraint_sets=p.constraint_sets)


def _run_sweep(robot, terrain, count, sample_idx, constraint_sets=None, stride=1):
    """One ragged batch of the first `count` enumerated candidates (combo x T x swing scale); every
    output element must be written (NaN pre-fill), the sampled candidates must match the oracle."""
    import torch

    from towr_amd import sweep

    model = ta.model_preset(robot, terrain)
    cands = sweep.enumerate_candidates(count * stride)[::stride]
    structs = sweep.candidate_structures(model, cands, constraint_sets=constraint_sets)
    assert len({(S.n, S.m, S.nnz) for S in structs}) > 1, "the sweep must be ragged"
    batch = ta.Batch(structs, list(range(count)), device=0)
    xs = _sweep_inputs(structs, model)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(np.concatenate(xs)).to(dev)
    g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device=dev)
    jac = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device=dev)
    batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all()), "constraint values left unwritten / non-finite"
    assert bool(torch.isfinite(jac).all()), "Jacobian values left unwritten / non-finite"
    gh, jh = g.cpu().numpy(), jac.cpu().numpy()

    def check(p):
        S = structs[p]
        rg, rp, ci, rj = _oracle_for(robot, terrain, S).eval(xs[p])   # (one oracle instance per candidate: not re-entrant)
        assert np.array_equal(rp, S.row_ptr) and np.array_equal(ci, S.col_idx), "pattern of candidate %d" % p
        assert_parity(S, *_split(batch, gh, jh, p), rg, rj, "%s/%s candidate %d %s" % (robot, terrain, p, cands[p]))

    if len(sample_idx) > 64:   # the whole sweep: oracle instances on a thread pool (ctypes drops the GIL in the calls)
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
            list(pool.map(check, sample_idx))
    else:
        for p in sample_idx:
            check(p)
    return cands


def _c5_samples():
    # lexicographic order: index = combo * 208 + i_T * 26 + j_scale.  Every combo at both ends of T and of the
    # swing scale, plus interior points.
    idx = []
    for combo in range(5):
        for iT, j in ((0, 0), (7, 25), (3, 13), (0, 25)):
            k = combo * 208 + iT * 26 + j
            if k < 1024:
                idx.append(k)
    return sorted(set(idx + [1023, 1000, 517]))


@pytest.mark.parametrize("robot", ["anymal", "go1"])
def test_sweep_c5_1024_stairs(robot):
    """BASELINE config 5: the 1024-candidate gait / phase-duration sweep on Stairs in ONE ragged batch
    (gait_generator.cc:54-105, quadruped_gait_generator.cc:76-87; Go1 constants go1_model.h:19-52); all 1024
    candidates are compared with the oracle."""
    idx = _c5_samples()
    assert len(idx) >= 16
    cands = _run_sweep(robot, "stairs", 1024, list(range(1024)))   # every candidate against the oracle
    assert {c[0] for c in (cands[i] for i in idx)} == {0, 1, 2, 3, 4}


def test_sweep_with_optimised_timings_ragged_multi_pass():
    """96 candidates spread over all five gait combos (every 10th of the enumeration), every constraint name of towr's
    list + optimised phase durations: 4800 dynamic passes and 2500 range-of-motion passes of DIFFERENT structures on 1024
    persistent workgroups, i.e. the hand-scheduled pipelines of dyn_phase_kernel / rom_phase_kernel carry records and x
    of one structure while computing another.  All candidates against the oracle."""
    cands = _run_sweep("anymal", "stairs", 96, list(range(96)), constraint_sets=127, stride=10)
    assert {c[0] for c in cands} == {0, 1, 2, 3, 4}


def test_layout_tables_are_shared_by_content():
    """twr_batch_create stores byte-identical layout tables of the dynamic set once (device_tables.h): candidates that
    differ in the total time only share all of them, candidates with different phase splits share none -- and sharing
    changes no bit of the results (same slices, same instruction sequence, other addresses)."""
    from towr_amd import sweep

    model = ta.model_preset("anymal", "stairs")
    # combo 1, swing scale 0.80, eight total times | combo 1, T = 2.0, six swing scales | one other combo
    cands = [(1, 1.2 + 0.2 * i, 0.80) for i in range(8)] + [(1, 2.0, 0.80 + 0.016 * j) for j in range(1, 7)] + [(3, 2.0, 0.9)]
    structs = sweep.candidate_structures(model, cands)
    order = list(range(len(cands)))[::-1] + [0, 3, 9]      # problems in another order than the structures, some twice
    batch = ta.Batch(structs, order, device=0)
    tb = batch.table_bytes()
    # 8 + 6 + 1 structures.  The eight T-siblings share what depends on the ee splines only (selectors, polynomial layout
    # records, tile records: ~12 KB of ~30 KB each); the base splines keep their fixed 0.1-s polynomials whatever T is, so
    # the variable offsets -- staging maps, base-polynomial offsets of the node records -- still differ.
    assert tb["dyn_layout_distinct"] <= 0.85 * tb["dyn_layout"] and tb["dyn_layout_distinct"] >= 0.5 * tb["dyn_layout"], tb
    assert tb["resident"] >= tb["dyn_layout"] > 0
    xs = _sweep_inputs([structs[s] for s in order], model)
    g, j = batch.eval_host(np.concatenate(xs))
    for p, s in enumerate(order):
        alone = ta.Batch([structs[s]], [0], device=0)     # nothing to share with
        assert alone.table_bytes()["dyn_layout_distinct"] == alone.table_bytes()["dyn_layout"]
        g1, j1 = alone.eval_host(xs[p])
        gd, jd = _split(batch, g, j, p)
        assert np.array_equal(gd, g1) and np.array_equal(jd, j1), "problem %d (structure %d)" % (p, s)
        if p in (0, 7, 14, 16):
            rg, _, _, rj = _oracle_for("anymal", "stairs", structs[s]).eval(xs[p])
            assert_parity(structs[s], gd, jd, rg, rj, "problem %d (structure %d)" % (p, s))
    # six different swing scales at one T: the phase splits differ, so which polynomial a time node falls into differs
    # (selectors, tiles, most staging maps) -- the polynomial layout records, which know nothing of times, still coincide
    six = ta.Batch(structs[8:14], list(range(6)), device=0).table_bytes()
    assert 0.5 * six["dyn_layout"] <= six["dyn_layout_distinct"] < six["dyn_layout"], six


def test_mid_size_sweep_streams_non_temporal_through_the_fused_launch():
    """384 enumerated candidates (every 2nd of the first 768: all gait combos of the enumeration's first four): 330 MB of
    output per evaluation and one structure per problem, so the batch streams its Jacobian values with non-temporal
    stores (twr_batch_streaming_stores), and ~6400 rom slices, so it goes through the FUSED launch -- the nt
    instantiation of eval_fused_kernel.  All candidates against the oracle.  Smaller or single-structure batches keep plain stores."""
    cands = _run_sweep("anymal", "stairs", 384, list(range(384)), stride=2)
    assert {c[0] for c in cands} == {0, 1, 2, 3}
    from towr_amd import sweep

    model = ta.model_preset("anymal", "stairs")
    structs = sweep.candidate_structures(model, sweep.enumerate_candidates(768)[::2])
    import torch

    batch = ta.Batch(structs, list(range(384)), device=0)
    assert batch.streaming_stores()
    # the same batch through the three separate kernels (per-kernel events on): the nt instantiations of dyn_kernel /
    # rom_kernel, bit for bit what the fused launch wrote
    x = torch.from_numpy(np.concatenate(_sweep_inputs(structs, model))).cuda()
    outs = []
    for profiled in (False, True):
        g = torch.full((int(batch.g_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
        j = torch.full((int(batch.jac_off[-1]),), float("nan"), dtype=torch.float64, device="cuda")
        if profiled:
            batch.profile_begin(1)
        batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if profiled:
            assert batch.profile_end()[1] == 1
        outs.append((g, j))
    assert bool(torch.isfinite(outs[0][1]).all()) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    del outs, x
    assert not ta.Batch(structs[:64], list(range(64)), device=0).streaming_stores()        # 55 MB of output
    assert not ta.Batch(structs[:1], [0] * 2048, device=0).streaming_stores()              # one structure for all problems


@pytest.mark.parametrize("terrain", ["gap", "stairs"])
def test_sweep_c4_64(terrain):
    """BASELINE config 4: batch of 64 enumerated candidate contact sequences on Gap / Stairs; all 64 checked."""
    _run_sweep("anymal", terrain, 64, list(range(64)))


def test_eval_check_flags_nonfinite_problems():
    """TWR_EVAL_CHECK (failure detection): per-problem NaN/Inf bits, other problems untouched."""
    import torch

    case = Case("anymal", "gap", ta.gait_combo(4, 1, 2.0))
    S = case.S
    xs = [case.x_perturbed(i, 2.0) for i in range(5)]
    xs[1][S.var_sets[1]["offset"] + 7] = float("nan")            # an Euler node value: poisons g and jac
    f0 = [v for v in S.var_sets if v["name"] == "ee-force_2"][0]
    xs[3][f0["offset"] + 4] = float("inf")                       # a force value: g only where the force enters linearly
    batch = ta.Batch([S], [0] * 5, device=0)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(np.concatenate(xs)).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    jac = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH | ta.EVAL_CHECK, st)
    status = batch.status(st)
    gh, jh = g.cpu().numpy(), jac.cpu().numpy()
    for p in range(5):
        want = (0 if np.isfinite(gh[batch.g_off[p]:batch.g_off[p + 1]]).all() else 1) | \
               (0 if np.isfinite(jh[batch.jac_off[p]:batch.jac_off[p + 1]]).all() else 2)
        assert status[p] == want, (p, status[p], want)
    assert status[0] == 0 and status[2] == 0 and status[4] == 0 and status[1] == 3 and status[3] & 1
    # values only: the Jacobian is neither written nor scanned
    batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_VALUES | ta.EVAL_CHECK, st)
    assert (batch.status(st) & 2).sum() == 0


def test_hostile_x_is_contained():
    """A line search can hand the callback anything.  Everything an index is derived from on the device goes through a
    clamp or a range check (locate_segment, the grid cell of a foothold), so NaN / Inf / 1e300 / negative or zero phase
    durations in some problems of a batch must neither fault nor hang, the poisoned problems are flagged by
    TWR_EVAL_CHECK and every other problem keeps the bits of a clean run."""
    import torch

    rng = np.random.default_rng(11)
    rough = (rng.uniform(-0.05, 0.3, size=(40, 30)).astype(np.float32), 0.05, (1.0, 0.1))
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for sets, terrain, kw in ((127, "grid_map", dict(grid=rough)), (127, "stairs", {}), (63, "grid_map", dict(grid=rough))):
        case = Case("anymal", terrain, ta.gait_combo(4, 1, 2.0), constraint_sets=sets, **kw)
        S = case.S
        B = 12
        clean = [case.x_perturbed(i, 1.6) for i in range(B)]
        xs = [x.copy() for x in clean]
        poison = [float("nan"), float("inf"), -float("inf"), 1e300, -1e300, 0.0, -0.3]
        bad = {}
        motion = [v for v in S.var_sets if v["name"].startswith("ee-motion")]
        sched = [v for v in S.var_sets if v["name"].startswith("ee-schedule")]
        for i, val in enumerate(poison):
            p = 1 + i
            if sched and i % 2 == 0:
                v = sched[i % len(sched)]
                xs[p][v["offset"] + i % v["size"]] = val                     # a phase duration
            elif val != 0.0 and val != -0.3:
                v = motion[i % len(motion)]
                xs[p][v["offset"]:v["offset"] + min(v["size"], 9)] = val       # foothold / swing node values
            else:
                continue
            bad[p] = val
        batch = ta.Batch([S], [0] * B, device=0)
        out = []
        for X in (clean, xs):
            x = torch.from_numpy(np.concatenate(X)).to(dev)
            g = torch.full((int(batch.g_off[-1]),), 7.0, dtype=torch.float64, device=dev)
            jac = torch.full((int(batch.jac_off[-1]),), 7.0, dtype=torch.float64, device=dev)
            batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH | ta.EVAL_CHECK, st)
            status = batch.status(st)
            out.append((g.cpu().numpy(), jac.cpu().numpy(), status))
        (g0, j0, s0), (g1, j1, s1) = out
        assert (s0 == 0).all()
        for p in range(B):
            a, b = _split(batch, g0, j0, p), _split(batch, g1, j1, p)
            if p in bad:
                if not np.isfinite(bad[p]):
                    assert s1[p] != 0, (sets, terrain, p, bad[p])
            else:
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and s1[p] == 0, (sets, terrain, p)


def test_grid_map_terrain():
    """`Grid` (grid_height_map.h:15-60), the terrain fpowr feeds the solver: float bilinear sample of the grid_map
    elevation layer, central-difference slopes, FLT_MAX outside -- terrain and force rows vs the oracle, with footholds
    in the interior, in the border band and outside the map (x_wild spreads them over [-0.5, 3] m)."""
    from tests.test_oracle_golden import _planar_grid_map

    rng = np.random.default_rng(5)
    rough = (rng.uniform(-0.05, 0.3, size=(48, 36)).astype(np.float32), 0.05, (1.0, 0.1))
    for gm, sets in ((_planar_grid_map(), 27), (rough, 63), (rough, 127)):
        case = Case("go1", "grid_map", ta.gait_combo(4, 1, 2.0), grid=gm, constraint_sets=sets)
        xs = [case.x_guess(1.6), case.x_perturbed(0, 1.6), case.x_perturbed(1, 1.2), case.x_wild(0), case.x_wild(1)]
        batch, g, j = _eval_case(case, xs)
        hit_outside = False
        for p, x in enumerate(xs):
            rg, _, _, rj = case.P.eval(x)
            assert_parity(case.S, *_split(batch, g, j, p), rg, rj, "grid_map sets %d x[%d]" % (sets, p), x=x)
            hit_outside |= bool((np.abs(rg) > 1e37).any())
        assert hit_outside, "the out-of-range branch (FLT_MAX) must be exercised"
    # two structures that share one map and one that has another: every distinct map is uploaded once
    other = (rough[0][::-1].copy(), 0.05, (1.0, 0.1))
    c1 = Case("anymal", "grid_map", ta.gait_combo(4, 0, 1.6), grid=rough)
    c2 = Case("anymal", "grid_map", ta.gait_combo(4, 2, 2.2), grid=c1.grid)
    c3 = Case("anymal", "grid_map", ta.gait_combo(4, 1, 2.0), grid=other)
    cases = [c1, c2, c3]
    batch = ta.Batch([c.S for c in cases], [0, 1, 2, 1], device=0)
    xs = [cases[s].x_perturbed(7 + i, 1.5) for i, s in enumerate([0, 1, 2, 1])]
    g, j = batch.eval_host(np.concatenate(xs))
    for p, s in enumerate([0, 1, 2, 1]):
        rg, _, _, rj = cases[s].P.eval(xs[p])
        assert_parity(cases[s].S, *_split(batch, g, j, p), rg, rj, "shared map problem %d" % p)


def test_grid_map_sweep_through_create_many():
    """A 64-candidate sweep on the `Grid` terrain fpowr runs on (footstep_plan_server.cc:155): all structures built by
    ONE threaded twr_structure_create_many_with_grid call, sharing one grid handle (uploaded once); every candidate equals
    the structure built one by one and the oracle on the same map."""
    from oracle import binding as ob
    from towr_amd import sweep

    rng = np.random.default_rng(21)
    elev = (0.04 * np.arange(60)[:, None] * 0.05 + rng.uniform(0.0, 0.05, size=(60, 44))).astype(np.float32)
    gm = ta.GridMap(elev, 0.05, (1.2, 0.0))
    model = ta.model_preset("anymal", "grid_map")
    cands = sweep.enumerate_candidates(64)
    structs = sweep.candidate_structures(model, cands, threads=4, grid=gm)
    assert len({(S.n, S.m, S.nnz) for S in structs}) > 1, "the sweep must be ragged"
    one = sweep.candidate_structure(model, cands[37], grid=gm)
    assert one.nnz == structs[37].nnz and np.array_equal(one.col_idx, structs[37].col_idx)
    batch = ta.Batch(structs, list(range(64)), device=0)
    xs = _sweep_inputs(structs, model)
    g, j = batch.eval_host(np.concatenate(xs))
    assert np.isfinite(j).all()
    for p in range(0, 64, 3):
        S = structs[p]
        prm = S.params
        P = ob.OracleProblem("anymal", "grid_map", S.schedule.durations(), S.schedule.contact(), dt_dynamic=prm.dt_dynamic,
                             dt_rom=prm.dt_rom, duration_base_poly=prm.duration_base_poly, polys_per_swing=prm.polys_per_swing,
                             polys_per_stance_force=prm.polys_per_stance_force, constraint_sets=prm.constraint_sets,
                             grid_map=(gm.elevation, gm.resolution, gm.position))
        rg, rp, ci, rj = P.eval(xs[p])
        assert np.array_equal(rp, S.row_ptr) and np.array_equal(ci, S.col_idx)
        assert_parity(S, *_split(batch, g, j, p), rg, rj, "grid_map sweep candidate %d" % p, x=xs[p])


def test_device_arg_min_matches_the_host_decision():
    """twr_batch_best (the planner's decision on the device) against towr_amd.dist.best_candidate on the same score table:
    NaN totals lose, the first index wins a tie, an all-NaN table answers 0; table sizes from one row to several blocks
    (the table may be longer than the batch: after an all-gather it holds every rank's candidates); replayed calls reuse
    the scratch."""
    import torch

    from towr_amd.dist import best_candidate

    case = Case("anymal", "flat", ta.gait_combo(4, 1, 2.0))
    batch = ta.Batch([case.S], [0, 0], device=0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    best = torch.zeros(2, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(5)
    for n in (1, 2, 63, 64, 65, 1000, 1024, 1025, 4097, 300001):
        for variant in ("plain", "ties", "nans", "all_nan", "inf"):
            t = rng.uniform(0.0, 10.0, size=(n, 16))
            if variant == "ties":
                t[:, :] = np.round(t, 0)      # many equal totals
            if variant == "nans":
                t[rng.uniform(size=n) < 0.3, 2] = np.nan
            if variant == "all_nan":
                t[:, 0] = np.nan
            if variant == "inf":
                t[:, 8] = np.inf
                t[n // 2, 8] = 1.0
            table = torch.from_numpy(t).to(dev)
            for fam in ((0, 1, 3, 4), (1,), (0, 1, 2, 3, 4, 5, 6, 7)):
                batch.best_device(table.data_ptr(), n, best.data_ptr(), families=fam, stream=st)
                got = best.cpu().numpy()
                idx, total = best_candidate(table, families=fam)
                assert int(got[0]) == idx and (got[1] == total or (np.isinf(got[1]) and np.isinf(total))), (n, variant, fam, got, idx, total)


def test_scores_of_a_structure_without_rows():
    """A constraint list that yields no rows at all (swing-* only, a gait without interior swing nodes): the evaluation writes
    nothing, the scoring kernel reads nothing (it must not touch g[-1]) and reports zero violation."""
    import torch

    case = random_case(5111)
    assert case.S.m == 0 and case.S.nnz == 0
    batch = ta.Batch([case.S], [0, 0, 0], device=0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(np.concatenate([case.x_wild(i) for i in range(3)])).to(dev)
    g = torch.zeros(1, dtype=torch.float64, device=dev)
    scores = torch.full((3, 16), -1.0, dtype=torch.float64, device=dev)
    best = torch.full((2,), -1.0, dtype=torch.float64, device=dev)
    batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, st)
    batch.score_best_device(g.data_ptr(), scores.data_ptr(), best.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert bool((scores == 0).all()) and best.cpu().tolist() == [0.0, 0.0]


def test_candidate_scores_and_contact_plans():
    """Sweep post-processing on the device: per-family bound violations (twr_batch_score) against numpy on the oracle's
    g and bounds, and fpowr::ExtractFootstepPlan (footstep_plan_extractor.h:69-133, minus the plane lookup) against
    the oracle restatement -- fixed and optimised timings, ragged batch."""
    import torch

    cases = [Case("anymal", "stairs", ta.gait_combo(4, 1, 2.0), constraint_sets=63),
             Case("anymal", "gap", ta.gait_combo(4, 0, 2.4, 0.9), constraint_sets=255, base_z_init=0.42),
             Case("go1", "flat", ta.gait_combo(4, 4, 1.8), constraint_sets=127),
             Case("anymal", "block", ta.gait_combo(4, 3, 2.2), constraint_sets=27),
             # more than 4096 rows (K = 460: 8.7 k): the scoring kernel takes sixteen rows per thread and trip
             Case("hyq", "slope", ta.gait_combo(4, 2, 2.0), constraint_sets=63, **k_params(2.0, 460))]
    assert cases[4].S.m > 2 * 4096
    order = [0, 1, 2, 3, 2, 0, 4]
    batch = ta.Batch([c.S for c in cases], order, device=0)
    xs = [cases[s].x_perturbed(i, 1.5) if i % 2 else cases[s].x_wild(i) for i, s in enumerate(order)]
    xs[4][cases[2].S.var_sets[0]["offset"] + 2] = float("nan")     # one problem with a poisoned base height
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(np.concatenate(xs)).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    scores = torch.full((len(order), 16), -1.0, dtype=torch.float64, device=dev)
    batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, st)
    batch.score_device(g.data_ptr(), scores.data_ptr(), st)
    torch.cuda.synchronize()
    sc = scores.cpu().numpy()
    # ... and the one-launch form: the same scores bit for bit, plus the shard's decision (twr_batch_score_best)
    from towr_amd.dist import best_candidate
    scores2 = torch.full((len(order), 16), -1.0, dtype=torch.float64, device=dev)
    best = torch.zeros(2, dtype=torch.float64, device=dev)
    for fam in ((0, 1, 3, 4), (1, 4)):
        batch.score_best_device(g.data_ptr(), scores2.data_ptr(), best.data_ptr(), families=fam, index_offset=1000, stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(scores2.cpu().numpy(), sc, equal_nan=True)
        idx, total = best_candidate(scores, families=fam)
        assert (int(best[0]), float(best[1])) == (1000 + idx, total)
    for p, s in enumerate(order):
        S = cases[s].S
        rg = cases[s].P.values(xs[p])
        lo, up = cases[s].P.bounds()
        viol = np.maximum(np.maximum(lo - rg, rg - up), 0.0)
        viol[np.isnan(rg)] = np.nan
        want = np.zeros((8, 2))
        for cs in S.con_sets:
            fam = [i for i, f in enumerate(ta.FAMILIES) if cs["name"].startswith(f)][0]
            v = viol[cs["offset"]:cs["offset"] + cs["size"]]
            want[fam, 0] = np.nan if (np.isnan(v).any() or np.isnan(want[fam, 0])) else max(want[fam, 0], v.max(initial=0.0))
            want[fam, 1] += v.sum()
        got = sc[p].reshape(8, 2)
        assert np.array_equal(np.isnan(got), np.isnan(want)), (p, got, want)
        ok = ~np.isnan(want)
        assert np.all(np.abs(got[ok] - want[ok]) <= 1e-9 * np.abs(want[ok]) + 1e-9), (p, got, want)
    assert np.isnan(sc[4]).any() and not np.isnan(sc[[0, 1, 2, 3, 5]]).any()
    # contact plans (dt = 0.01 and the time horizon of fpowr, footstep_plan_server.cc:31)
    for dt, horizon in ((0.01, 2.0), (0.037, 3.0)):
        max_steps = max(c.S.contact_steps_max() for c in cases)
        n_ee = 4
        rec = 2 + 4 * n_ee
        out = torch.full((len(order), max_steps, rec), float("nan"), dtype=torch.float64, device=dev)
        counts = torch.zeros(len(order), dtype=torch.int32, device=dev)
        batch.contact_plan_device(x.data_ptr(), dt, horizon, out.data_ptr(), max_steps, counts.data_ptr(), st)
        torch.cuda.synchronize()
        oh, ch = out.cpu().numpy(), counts.cpu().numpy()
        for p, s in enumerate(order):
            if p == 4:
                continue   # (NaN durations in x: locate_segment is undefined there, like the reference's assert)
            ref = cases[s].P.contact_plan(xs[p], dt, horizon)
            assert ch[p] == ref.shape[0] and 2 <= ch[p] <= max_steps, (p, ch[p], ref.shape)
            got = oh[p, :ch[p]]
            assert np.array_equal(got[:, 2:2 + n_ee], ref[:, 2:2 + n_ee])          # contact flags
            assert np.abs(got[:, :2] - ref[:, :2]).max() <= 1e-12                    # t, duration
            assert np.abs(got[:, 2 + n_ee:] - ref[:, 2 + n_ee:]).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        # a caller-chosen max_steps smaller than the number of footstep states: counts report what was found, the first
        # max_steps records are the same as before -- durations included (the last kept record ends at the first dropped state)
        few = 3
        out3 = torch.full((len(order), few, rec), float("nan"), dtype=torch.float64, device=dev)
        counts3 = torch.zeros(len(order), dtype=torch.int32, device=dev)
        batch.contact_plan_device(x.data_ptr(), dt, horizon, out3.data_ptr(), few, counts3.data_ptr(), st)
        torch.cuda.synchronize()
        o3, c3 = out3.cpu().numpy(), counts3.cpu().numpy()
        for p in range(len(order)):
            if p == 4:
                continue
            assert c3[p] == ch[p]
            k = min(few, ch[p])
            assert np.array_equal(o3[p, :k], oh[p, :k]), (p, o3[p, :k, :2], oh[p, :k, :2])
        # nearest planar region of every foot in contact (fpowr NearestPlaneLookup): rotated, overlapping, open and
        # closed polygons along the path of the robot
        from oracle import binding as ob
        rng = np.random.default_rng(11)
        regions, boundaries = [], []
        for r in range(14):
            yaw, k = rng.uniform(-np.pi, np.pi), int(rng.integers(3, 9))
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            pts = np.stack([np.cos(ang), np.sin(ang)], axis=1) * rng.uniform(0.15, 0.6, (k, 1))
            if r % 2 == 0:
                pts = np.concatenate([pts, pts[:1]])          # closed ring (first point repeated); the others stay open
            regions.append([rng.uniform(-0.5, 2.5), rng.uniform(-0.6, 0.6), rng.uniform(0, 0.3), 0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
            boundaries.append(pts)
        planes = ta.Planes(regions, boundaries, device=0)
        start = planes.start
        wxy = ob.planes_world_xy(regions, np.concatenate(boundaries), start)
        assert np.allclose(planes.world_xy, wxy, rtol=0, atol=1e-15)
        idx = torch.full((len(order), max_steps, n_ee), -7, dtype=torch.int32, device=dev)
        batch.contact_planes_device(planes, out.data_ptr(), counts.data_ptr(), max_steps, idx.data_ptr(), st)
        torch.cuda.synchronize()
        ih = idx.cpu().numpy()
        seen = set()
        for p, s in enumerate(order):
            if p == 4:
                continue
            for k in range(max_steps):
                for e in range(n_ee):
                    if k < ch[p] and oh[p, k, 2 + e] != 0.0:
                        want = ob.nearest_plane(planes.world_xy, start, oh[p, k, 2 + n_ee + 3 * e], oh[p, k, 2 + n_ee + 3 * e + 1])
                        seen.add(want)
                    else:
                        want = -1
                    assert ih[p, k, e] == want, (p, k, e, ih[p, k, e], want)
        assert len(seen - {-1}) >= 3
        none = ta.Planes(np.zeros((0, 7)), [], device=0)          # a terrain message without regions: -1 everywhere
        batch.contact_planes_device(none, out.data_ptr(), counts.data_ptr(), max_steps, idx.data_ptr(), st)
        torch.cuda.synchronize()
        assert (idx.cpu().numpy() == -1).all()


def test_eval_is_graph_capturable_and_stream_ordered():
    """twr_batch_eval is asynchronous on the caller's stream and capturable in a hipGraph (include/towr_amd.h): capture one
    callback, replay it on new x, compare with an eager evaluation; two batches on two streams run concurrently."""
    import torch

    case = Case("anymal", "stairs", ta.gait_combo(4, 1, 2.0), constraint_sets=63)
    S = case.S
    B = 64
    batch = ta.Batch([S], [0] * B, device=0)
    dev = torch.device("cuda", 0)
    xs = np.stack([case.x_perturbed(i, 1.5) for i in range(B)])
    x = torch.from_numpy(xs.reshape(-1)).to(dev)
    g = torch.zeros(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    jac = torch.zeros(int(batch.jac_off[-1]), dtype=torch.float64, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # warm-up outside the capture (module load)
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager_g, eager_j = g.clone(), jac.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
    g.zero_()
    jac.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g, eager_g) and torch.equal(jac, eager_j)
    # new inputs through the same graph
    xs2 = np.stack([case.x_perturbed(100 + i, 1.5) for i in range(B)])
    x.copy_(torch.from_numpy(xs2.reshape(-1)))
    graph.replay()
    torch.cuda.synchronize()
    rg, _, _, rj = case.P.eval(xs2[17])
    assert_parity(S, g.cpu().numpy()[batch.g_off[17]:batch.g_off[18]], jac.cpu().numpy()[batch.jac_off[17]:batch.jac_off[18]],
                  rg, rj, "graph replay", x=xs2[17])
    # optimised timings: pre-pass + the dynamic-LDS kernels inside a graph, incl. a structure whose images exceed 64 KB
    # (the LDS limit of those instantiations is raised at batch creation, not at launch)
    big = Case("biped", "gap", _many_phases(2, 31, 2), constraint_sets=127)
    small = Case("biped", "stairs", ta.gait_combo(2, 0, 2.0), constraint_sets=127)
    tb = ta.Batch([big.S, small.S], [0, 1, 1, 0], device=0)
    txs = [c.x_perturbed(30 + i, 1.2) for i, c in enumerate([big, small, small, big])]
    tx = torch.from_numpy(np.concatenate(txs)).to(dev)
    tg = torch.zeros(int(tb.g_off[-1]), dtype=torch.float64, device=dev)
    tj = torch.zeros(int(tb.jac_off[-1]), dtype=torch.float64, device=dev)
    with torch.cuda.stream(side):
        tb.eval_device(tx.data_ptr(), tg.data_ptr(), tj.data_ptr(), ta.EVAL_BOTH, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eg, ej = tg.clone(), tj.clone()
    tgraph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(tgraph):
        tb.eval_device(tx.data_ptr(), tg.data_ptr(), tj.data_ptr(), ta.EVAL_BOTH, torch.cuda.current_stream().cuda_stream)
    tg.zero_()
    tj.zero_()
    tgraph.replay()
    torch.cuda.synchronize()
    assert torch.equal(tg, eg) and torch.equal(tj, ej)
    rg, _, _, rj = big.P.eval(txs[3])
    assert_parity(big.S, tg.cpu().numpy()[tb.g_off[3]:tb.g_off[4]], tj.cpu().numpy()[tb.jac_off[3]:tb.jac_off[4]], rg, rj,
                  "graph replay, optimised timings", x=txs[3])
    # two batches, two streams, in flight together
    batch2 = ta.Batch([S], [0] * B, device=0)
    g2, j2 = torch.zeros_like(g), torch.zeros_like(jac)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    s2.wait_stream(torch.cuda.current_stream())
    g.zero_(); jac.zero_()
    torch.cuda.synchronize()
    for _ in range(3):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, s1.cuda_stream)
        batch2.eval_device(x.data_ptr(), g2.data_ptr(), j2.data_ptr(), ta.EVAL_BOTH, s2.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(g, g2) and torch.equal(jac, j2)


def test_bench_prints_one_contract_line():
    """`python bench.py` as the driver runs it (N = 1; fewer steps): exactly one JSON line on stdout with the contract's keys,
    the `roofline` and `cpu_baseline` objects and the extra legs of the default run."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1"], capture_output=True,
                       text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "callbacks/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["problems_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.2 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * rf["achieved"]
    assert rf["traffic"] is None or 0.9 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.2
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 1.0 and cb["unit"] == "callbacks/s" and cb["sample"]
    for leg in ("timings_c3", "all_sets_c3", "scale_c5"):
        assert d[leg]["value"] > 0 and d[leg]["unit"] == "callbacks/s", leg
    assert d["scale_c5"]["candidates"] == 1024 and d["scale_c5"]["world_size"] == 1
