"""Host logic of the product library vs the oracle: variable layout, CSR pattern, bounds, initial
guess, gait generator; and the C ABI surface (symbols, error behaviour).  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import towr_amd as ta
from oracle import binding as ob
from tests.common import Case, hopper_schedule, k_params, random_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = [
    ("monoped", "flat", None, 2.0, {}),
    ("monoped", "block", 3, 1.7, {}),
    ("biped", "flat", 0, 2.0, k_params(2.0, 100)),
    ("anymal", "flat", 1, 2.0, k_params(2.0, 200)),
    ("anymal", "gap", 0, 2.0, {}),
    ("anymal", "stairs", 2, 2.4, {}),
    ("biped", "stairs", 1, 1.8, {}),
    ("biped", "slope", 3, 2.4, {}),
    ("hyq", "chimney_lr", 4, 2.3, dict(polys_per_swing=3, polys_per_stance_force=2, duration_base_poly=0.15)),
    ("go1", "stairs", 0, 2.3, dict(polys_per_swing=1, polys_per_stance_force=1, duration_base_poly=0.07,
                                   dt_dynamic=0.033, dt_rom=0.05)),
    ("go1", "chimney", 3, 2.6, dict(dt_dynamic=0.25, dt_rom=0.3)),
    # towr's whole default constraint list (parameters.cc:55-60) and odd subsets of it
    ("monoped", "flat", None, 2.0, dict(constraint_sets=63)),
    ("biped", "block", 0, 2.0, dict(constraint_sets=63)),
    ("anymal", "flat", 1, 2.0, dict(constraint_sets=63, **k_params(2.0, 200))),
    ("hyq", "gap", 2, 2.1, dict(constraint_sets=63, polys_per_swing=3, duration_base_poly=0.13)),
    ("go1", "slope", 0, 2.3, dict(constraint_sets=63, polys_per_swing=1)),
    ("anymal", "stairs", 0, 2.4, dict(constraint_sets=2 | 32)),
    ("biped", "flat", 1, 1.8, dict(constraint_sets=4)),
    ("anymal", "chimney", 1, 2.0, dict(constraint_sets=1 | 16 | 32)),
    # optimised phase durations (Parameters::OptimizePhaseDurations): ee-schedule sets, all-variables rows
    ("monoped", "flat", None, 2.0, dict(constraint_sets=127)),
    ("biped", "stairs", 0, 2.0, dict(constraint_sets=127)),
    ("anymal", "gap", 1, 2.0, dict(constraint_sets=127)),
    ("anymal", "flat", 1, 2.0, dict(constraint_sets=127, **k_params(2.0, 200))),
    ("hyq", "slope", 3, 2.2, dict(constraint_sets=127, polys_per_swing=3, polys_per_stance_force=2)),
    ("go1", "block", 2, 1.9, dict(constraint_sets=2 | 64)),
    ("biped", "flat", 1, 1.8, dict(constraint_sets=8 | 64)),
    # baseMotion (Parameters::BaseRom), alone and with everything else
    ("monoped", "flat", None, 2.0, dict(constraint_sets=128, base_z_init=0.58)),
    ("anymal", "stairs", 1, 2.0, dict(constraint_sets=255, base_z_init=0.42, dt_base_motion=0.031)),
    ("biped", "gap", 0, 2.0, dict(constraint_sets=63 | 128, base_z_init=0.65)),
]


def make(robot, terrain, combo, T, kw):
    n_ee = ta.model_preset(robot, terrain).n_ee
    sched = hopper_schedule() if combo is None else ta.gait_combo(n_ee, combo, T)
    return Case(robot, terrain, sched, **kw)


@pytest.mark.parametrize("cfg", CONFIGS, ids=["%s-%s-c%s-s%d" % (c[0], c[1], c[2], c[4].get("constraint_sets", 27)) for c in CONFIGS])
def test_structure_matches_oracle(cfg):
    case = make(*cfg)
    S, P = case.S, case.P
    assert (S.n, S.m, S.nnz) == (P.n, P.m, P.nnz)
    assert [(s["name"], s["size"]) for s in S.var_sets] == P.var_sets
    assert [(s["name"], s["size"]) for s in S.con_sets] == P.con_sets
    x = np.random.default_rng(0).normal(size=P.n)
    _, rp, ci, _ = P.eval(x)
    assert np.array_equal(rp, S.row_ptr) and np.array_equal(ci, S.col_idx)
    # the pattern must not depend on x (Ipopt fixes the structure once)
    _, rp2, ci2, _ = P.eval(np.zeros(P.n))
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    lo, up = S.bounds()
    lo2, up2 = P.bounds()
    assert np.array_equal(lo, lo2) and np.array_equal(up, up2)
    # set offsets are consistent with the CSR arrays
    for s in S.con_sets:
        assert s["nnz_offset"] == S.row_ptr[s["offset"]]
        assert s["nnz"] == S.row_ptr[s["offset"] + s["size"]] - S.row_ptr[s["offset"]]
    ee0 = np.array([[0.1 * i, 0.05 * i, 0.0] for i in range(S.n_ee)])
    args = ([0.1, 0.2, 0.5], [0.01, 0.02, 0.3], [1.3, 0.1, 0.5], [0.0, 0.0, 0.7], ee0)
    assert np.array_equal(S.initial_guess(*args), P.initial_guess(*args))
    # variable bounds: start state and final base state (nlp_formulation.cc:109-122,151)
    rng = np.random.default_rng(3)
    vb = (rng.normal(size=12), rng.normal(size=12), ee0)
    lo, up = S.variable_bounds(*vb)
    lo2, up2 = P.variable_bounds(*vb)
    assert np.array_equal(lo, lo2) and np.array_equal(up, up2)
    fixed = lo == up
    n_fixed_ee = sum(3 for _ in range(S.n_ee))
    n_sched = sum(v["size"] for v in S.var_sets if v["name"].startswith("ee-schedule"))
    free = ~fixed
    free[S.n - n_sched:] = False
    assert fixed.sum() == 12 + 11 + n_fixed_ee and np.all(lo[free] == -1e20) and np.all(up[free] == 1e20)
    assert np.all(lo[S.n - n_sched:] == 0.2) and np.all(up[S.n - n_sched:] == 1.0)


@pytest.mark.parametrize("seed", range(40))
def test_random_structures_match_oracle(seed):
    """Seeded fuzz over robots, terrains, phase counts, polynomial counts, time steps and set masks."""
    case = random_case(seed)
    S, P = case.S, case.P
    assert (S.n, S.m, S.nnz) == (P.n, P.m, P.nnz)
    assert [(s["name"], s["size"]) for s in S.con_sets] == P.con_sets
    assert [(s["name"], s["size"]) for s in S.var_sets] == P.var_sets
    _, rp, ci, _ = P.eval(case.x_wild(seed))
    assert np.array_equal(rp, S.row_ptr) and np.array_equal(ci, S.col_idx)
    lo, up = S.bounds()
    lo2, up2 = P.bounds()
    assert np.array_equal(lo, lo2) and np.array_equal(up, up2)


@pytest.mark.parametrize("n_ee", [1, 2, 4])
@pytest.mark.parametrize("combo", range(5))
def test_gait_generator_matches_oracle_tables(n_ee, combo):
    for T in (1.3, 2.0, 3.1):
        s = ta.gait_combo(n_ee, combo, T)
        pd, con = ob.gait(n_ee, combo, T)
        assert s.contact() == con
        for e in range(n_ee):
            assert np.array_equal(np.array(s.durations()[e]), pd[e])
            assert abs(sum(s.durations()[e]) - T) < 1e-12


def test_gait_known_answer_flying_trot():
    """quadruped_gait_generator.cc:80-81,113-126,223-255 by hand: Stand(0.3) + 3x Run2 + Run2E + Stand."""
    s = ta.gait_combo(4, 1, 3.0)  # table total = 0.3 + 3*1.0 + 0.4 + 0.3 = 4.0
    k = 3.0 / 4.0
    lf = [0.3 + 0.4, 0.1 + 0.4 + 0.1, 0.4, 0.6, 0.4, 0.6, 0.4 + 0.3]
    rf = [0.3, 0.4 + 0.1, 0.4, 0.6, 0.4, 0.6, 0.4, 0.1 + 0.4, 0.3]
    assert np.allclose(s.durations()[0], np.array(lf) * k, rtol=1e-14)
    assert np.allclose(s.durations()[1], np.array(rf) * k, rtol=1e-14)
    assert np.allclose(s.durations()[2], s.durations()[1], rtol=0) and np.allclose(s.durations()[3], s.durations()[0], rtol=0)
    # swing_scale only stretches phases with a foot in the air, then renormalises to T
    s2 = ta.gait_combo(4, 1, 3.0, 0.8)
    assert abs(sum(s2.durations()[0]) - 3.0) < 1e-12 and s2.durations()[0][1] < s.durations()[0][1]


def test_robot_presets_match_reference_constants():
    m = ta.model_preset("anymal", "gap")  # anymal_model.h:44-67
    assert (m.n_ee, m.mass, m.gravity, m.friction, m.force_limit) == (4, 29.5, 9.80665, 0.5, 1000.0)
    assert list(m.inertia) == [0.946438, 1.94478, 2.01835, 0.000938112, -0.00595386, -0.00146328]
    assert [list(m.nominal_stance[e]) for e in range(4)] == [[0.34, 0.19, -0.42], [0.34, -0.19, -0.42],
                                                             [-0.34, 0.19, -0.42], [-0.34, -0.19, -0.42]]
    assert list(m.max_dev) == [0.15, 0.1, 0.10]
    g = ta.model_preset("go1", "flat")  # go1_model.h:19-52
    assert g.mass == 12.84 and g.nominal_stance[0][1] == 0.04675 + 0.08
    p = ta.params_default()  # parameters.cc:43-50
    assert (p.dt_dynamic, p.dt_rom, p.duration_base_poly, p.polys_per_swing, p.polys_per_stance_force) == (0.1, 0.08, 0.1, 2, 3)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "towr_amd.h")).read()
    names = set(re.findall(r"\b(twr_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 18
    lib = C.CDLL(ta.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts_match_the_header():
    """ctypes mirrors must have the sizes the C compiler gives the header's structs."""
    import subprocess
    import tempfile

    src = '#include "towr_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu",sizeof(twr_model),sizeof(twr_schedule),sizeof(twr_params),sizeof(twr_sizes),sizeof(twr_set_info));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    assert sizes == [C.sizeof(ta.Model), C.sizeof(ta.Schedule), C.sizeof(ta.Params), C.sizeof(ta.Sizes), C.sizeof(ta.SetInfo)]


def test_error_behaviour():
    L = ta.lib()
    m = ta.Model()
    assert L.twr_model_preset(99, 0, C.byref(m)) == -1 and b"robot" in L.twr_last_error()
    assert L.twr_model_preset(0, 42, C.byref(m)) == -1 and b"terrain" in L.twr_last_error()
    s = ta.Schedule()
    assert L.twr_gait_combo(3, 0, 2.0, 1.0, C.byref(s)) == -1   # no 3-legged gait generator (gait_generator.cc:43-52)
    assert L.twr_gait_combo(4, 7, 2.0, 1.0, C.byref(s)) == -1
    assert L.twr_gait_combo(4, 1, -1.0, 1.0, C.byref(s)) == -1
    # feet whose phase durations do not sum to the same T (assert in parameters.cc:120-123)
    bad = ta.schedule([[0.5, 0.5], [0.5, 0.6]], [1, 1])
    with pytest.raises(ta.TowrError, match="same T"):
        ta.Structure(ta.model_preset("biped", "flat"), bad)
    with pytest.raises(ta.TowrError, match="n_ee"):
        ta.Structure(ta.model_preset("anymal", "flat"), bad)
    with pytest.raises(ta.TowrError):
        ta.Structure(ta.model_preset("biped", "flat"), ta.gait_combo(2, 0, 2.0), ta.params_default(dt_dynamic=0.0))
    for mask in (0, 256, -1):
        with pytest.raises(ta.TowrError, match="constraint_sets"):
            ta.Structure(ta.model_preset("biped", "flat"), ta.gait_combo(2, 0, 2.0), ta.params_default(constraint_sets=mask))
    # SwingConstraint "assumes ... starting and ending in stance" (swing_constraint.cc:66): a foot that
    # starts in swing has no previous node; refused at build time (the reference would throw in at())
    flying = ta.schedule([[0.3, 0.5, 0.3, 0.5]], [0])
    ta.Structure(ta.model_preset("monoped", "flat"), flying)  # fine without the swing set
    with pytest.raises(ta.TowrError, match="stance"):
        ta.Structure(ta.model_preset("monoped", "flat"), flying, ta.params_default(constraint_sets=63))
    h = C.c_void_p()
    assert L.twr_structure_create(None, None, None, C.byref(h)) == -1
    assert L.twr_batch_eval(None, None, None, None, 3, None) == -1


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a GPU the product path must fail loudly, not fall back to anything."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    S = ta.Structure(ta.model_preset("monoped", "flat"), hopper_schedule())
    with pytest.raises(ta.TowrError, match="no HIP device|TWR|hip"):
        ta.Batch([S], [0])


def test_sample_counts_match_oracle():
    """fpowr::GetTrajectory: samples while t <= T + 1e-5 with t accumulated (footstep_plan_extractor.h:27-50)."""
    for robot, combo, T, dt in [("anymal", 1, 2.0, 0.01), ("biped", 0, 1.7, 0.013), ("monoped", 2, 0.93, 0.1), ("hyq", 4, 2.4, 0.007)]:
        n_ee = ta.model_preset(robot, "flat").n_ee
        case = Case(robot, "flat", ta.gait_combo(n_ee, combo, T))
        ref = case.P.sample_trajectory(case.x_guess(), dt)
        assert case.S.sample_count(dt) == ref.shape[0]
        assert ref.shape[1] == 20 + 13 * n_ee and ref[0, 0] == 0.0 and abs(ref[-1, 0] - T) <= dt + 1e-5
    with pytest.raises(ta.TowrError):
        case.S.sample_count(0.0)


def test_sweep_enumeration_and_sharding():
    from towr_amd import sweep

    c = sweep.enumerate_candidates(1024)
    assert len(c) == 1024 and c[0] == (0, 1.2, 0.80) and c[26][1] == pytest.approx(1.4) and c[208][0] == 1
    with pytest.raises(ValueError):
        sweep.enumerate_candidates(2000)
    m = ta.model_preset("anymal", "stairs")
    sizes = [sweep.candidate_structure(m, x).algorithmic_bytes for x in c[:64:4]]
    assert len(set(sizes)) > 1          # ragged
    S = sweep.candidate_structure(m, c[40])
    assert S.k_dynamic == 200 and S.k_rom == 200
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 4, 8):
        w = rng.integers(800_000, 900_000, size=1024)
        b = sweep.shard_bounds(w, world)
        assert b[0] == 0 and b[-1] == 1024 and all(b[i] <= b[i + 1] for i in range(world))
        loads = [w[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(loads) - min(loads) <= 2 * w.max()
    assert sweep.shard_bounds([5, 1, 1, 1, 1, 1], 2) == [0, 1, 6]
    with pytest.raises(ValueError):      # more ranks than candidates: every rank raises the same error
        sweep.shard_bounds([1, 1], 4)
    assert sweep.shard_bounds([9, 1, 1], 3) == [0, 1, 2, 3]   # never an empty shard
    assert sweep.shard_bounds([1, 1, 50], 3) == [0, 1, 2, 3]
    # the weight is BYTES per callback.  twr_candidate_bytes gives 8 (n + m + nnz) exactly, without building
    # the device tables; a sweep that mixes time-node counts is then balanced by bytes, not by candidate count
    sub = c[:64:4]
    assert sweep.candidate_bytes(m, sub, threads=2).tolist() == sizes
    mixed_k = [200] * 8 + [100] * 8
    wb = sweep.candidate_bytes(m, c[:16], k_nodes=mixed_k, threads=2)
    assert wb[:8].min() > 1.9 * wb[8:].max()
    b = sweep.shard_bounds(wb, 2)
    assert b[1] in (5, 6)       # about two thirds of the bytes sit in the first eight candidates
    with pytest.raises(ta.TowrError):
        sweep.candidate_bytes(m, c[:2], constraint_sets=0)


def test_create_many_equals_one_by_one():
    """twr_structure_create_many (threaded sweep setup) builds the same structures as twr_structure_create."""
    from towr_amd import sweep

    m = ta.model_preset("anymal", "stairs")
    cands = sweep.enumerate_candidates(1040)[::37]
    many = sweep.candidate_structures(m, cands, threads=4)
    for c, S in zip(cands, many):
        one = sweep.candidate_structure(m, c)
        assert (S.n, S.m, S.nnz) == (one.n, one.m, one.nnz)
        assert np.array_equal(S.row_ptr, one.row_ptr) and np.array_equal(S.col_idx, one.col_idx)
        assert S.con_sets == one.con_sets and S.var_sets == one.var_sets
    # a failing candidate fails the whole call and leaves nothing behind
    bad = ta.params_default(constraint_sets=0)
    with pytest.raises(ta.TowrError, match="structure 1"):
        ta.Structure.create_many(m, [many[0].schedule, many[1].schedule], [many[0].params, bad], threads=2)


def test_create_many_with_a_shared_grid_and_grid_info():
    """twr_structure_create_many_with_grid: every structure of a sweep over a gridded terrain shares one grid handle;
    a gridded terrain id without a grid (or with the other kind of grid) is rejected; twr_terrain_grid_info returns what
    a rank needs to rebuild the handle (dist.broadcast_grid)."""
    import ctypes as C

    from towr_amd import sweep

    elev = (np.arange(30 * 22, dtype=np.float32).reshape(30, 22) % 13) * 0.01
    gm = ta.GridMap(elev, 0.05, (0.8, -0.2))
    csv = ta.TerrainGrid(np.arange(6 * 9, dtype=np.float64).reshape(6, 9) * 0.02)
    m = ta.model_preset("anymal", "grid_map")
    cands = sweep.enumerate_candidates(1040)[::97]
    many = sweep.candidate_structures(m, cands, threads=3, grid=gm)
    for c, S in zip(cands, many):
        one = sweep.candidate_structure(m, c, grid=gm)
        assert (S.n, S.m, S.nnz) == (one.n, one.m, one.nnz) and np.array_equal(S.col_idx, one.col_idx)
    with pytest.raises(ta.TowrError, match="grid"):
        sweep.candidate_structures(m, cands[:2], threads=2)               # gridded terrain id, no grid
    with pytest.raises(ta.TowrError, match="kind"):
        sweep.candidate_structures(m, cands[:2], threads=2, grid=csv)     # CSV heights for a grid_map terrain
    for handle, kind, shape, res, pos, dtype in ((gm, 1, (30, 22), 0.05, (0.8, -0.2), np.float32), (csv, 0, (6, 9), None, None, np.float64)):
        k, n0, n1 = C.c_int32(), C.c_int32(), C.c_int32()
        r, px, py = C.c_double(), C.c_double(), C.c_double()
        data = C.c_void_p()
        ta._check(ta.lib().twr_terrain_grid_info(handle._h, C.byref(k), C.byref(n0), C.byref(n1), C.byref(r), C.byref(px), C.byref(py),
                                                 C.byref(data)))
        assert (k.value, n0.value, n1.value) == (kind,) + shape
        got = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_float if kind else C.c_double)), shape=(shape[0] * shape[1],))
        want = handle.elevation.reshape(-1, order="F") if kind else handle.heights.reshape(-1)
        assert np.array_equal(got, want)
        if kind:
            assert (r.value, px.value, py.value) == (res,) + pos


def test_base_motion_needs_the_initial_base_height():
    """twr_params_default leaves base_z_init unset (NaN): enabling baseMotion without it is rejected instead of
    silently producing the infeasible bounds [-0.02, 0.1] (base_motion_constraint.cc:51-55)."""
    m = ta.model_preset("anymal", "flat")
    with pytest.raises(ta.TowrError, match="base_z_init"):
        ta.Structure(m, ta.gait_combo(4, 1, 2.0), ta.params_default(constraint_sets=ta.SETS_EVERY))
    S = ta.Structure(m, ta.gait_combo(4, 1, 2.0), ta.params_default(constraint_sets=ta.SETS_EVERY, base_z_init=0.42))
    lo, up = S.bounds()
    bm = [s for s in S.con_sets if s["name"] == "baseMotion"][0]
    assert lo[bm["offset"] + 5] == pytest.approx(0.40) and up[bm["offset"] + 5] == pytest.approx(0.52)


def test_place_outputs_keeps_the_fastest_allocation():
    """towr_amd.placement.place_outputs (what bench.py places its x / g / Jacobian buffers with): allocates `tries` times, each
    time behind a ballast allocation that is freed again, times a few steps on each and keeps the fastest; tries = 1 takes the
    first allocation untimed.  Driven here with stand-ins for torch and for the buffers (no device)."""
    import time
    import types

    from towr_amd.placement import PLACEMENT_BALLAST_GB, place_outputs

    log = []
    fake = types.SimpleNamespace(float64="f64", cuda=types.SimpleNamespace(synchronize=lambda: None, empty_cache=lambda: log.append("empty_cache"),
                                                                      mem_get_info=lambda dev: (200 << 30, 288 << 30)),
                                 empty=lambda n, dtype=None, device=None: log.append(("ballast", n)) or object())
    cost = {0: 0.004, 1: 0.001, 2: 0.003}   # seconds per step of allocation i
    made = []

    def alloc():
        made.append(len(made))
        return ("buffers", made[-1])

    def run_steps(bufs, n):
        time.sleep(cost[bufs[1]] * n)

    kept, report = place_outputs(fake, "dev", alloc, run_steps, 3)
    assert kept == ("buffers", 1) and report["kept"] == 1 and len(report["tries"]) == 3
    assert [t["ballast_GB"] for t in report["tries"]] == list(PLACEMENT_BALLAST_GB[:3])
    assert report["tries"][1]["ms_per_step"] < report["tries"][2]["ms_per_step"] < report["tries"][0]["ms_per_step"]
    assert [e for e in log if isinstance(e, tuple)] == [("ballast", int(g * (1 << 27))) for g in PLACEMENT_BALLAST_GB[1:3]]
    kept, report = place_outputs(fake, "dev", alloc, run_steps, 1)
    assert kept == ("buffers", 3) and report["tries"] == [{"ballast_GB": 0.0, "ms_per_step": None}]
    # a device that is nearly full: the second placement would not fit beside the first -- the first is kept, and the line says why
    state = {"free": 100 << 30}

    def alloc_big():
        state["free"] -= 60 << 30
        return ("buffers", 0)

    fake.cuda.mem_get_info = lambda dev: (state["free"], 288 << 30)
    kept, report = place_outputs(fake, "dev", alloc_big, run_steps, 4)
    assert kept == ("buffers", 0) and len(report["tries"]) == 2 and "skipped" in report["tries"][1]


def test_place_in_arena_keeps_the_fastest_window():
    """towr_amd.placement.place_outputs with jac_numel: the Jacobian buffer is a window of ONE arena allocation (0.7 of the free
    device memory), 2 x tries windows spread evenly over it, the fastest kept; without an arena (no mem_get_info) it falls back
    to placement by allocation.  Stand-ins for torch and the buffers (no device)."""
    import time
    import types

    from towr_amd import placement

    class Arena:
        def __init__(self, numel):
            self.n = numel

        def numel(self):
            return self.n

        def __getitem__(self, sl):
            return ("window", sl.start, sl.stop - sl.start)

    made = []
    fake = types.SimpleNamespace(float64="f64", empty=lambda n, dtype=None, device=None: made.append(n) or Arena(n),
                                 cuda=types.SimpleNamespace(synchronize=lambda: None, empty_cache=lambda: None,
                                                            mem_get_info=lambda dev: (100 << 30, 288 << 30)))
    placement._ARENAS.pop("fake-dev", None)
    n_jac = 1 << 27   # 1 GiB of doubles

    def alloc(jac=None):
        assert jac is not None
        return ("x", "g", jac)

    def run_steps(bufs, n):   # windows in the third quarter of the arena are the fast ones
        frac = bufs[2][1] / made[0]
        time.sleep((0.001 if 0.5 < frac < 0.75 else 0.003) * n)

    (x, g, jac), report = placement.place_outputs(fake, "fake-dev", alloc, run_steps, 4, jac_numel=n_jac)
    assert made == [int(0.7 * (100 << 30)) // (1 << 21) * (1 << 18)] and report["arena_GB"] == 70.0
    assert len(report["tries"]) == 8 and jac[2] == n_jac and jac[1] % 32 == 0 and 0.5 < jac[1] / made[0] < 0.75
    assert report["tries"][report["kept"]]["ms_per_step"] == min(t["ms_per_step"] for t in report["tries"])
    # a second call reuses the arena of the process
    placement.place_outputs(fake, "fake-dev", alloc, run_steps, 2, jac_numel=n_jac)
    assert len(made) == 1
    placement._ARENAS.pop("fake-dev", None)


def test_values_only_items_are_cut_at_64_nodes_and_eight_polynomials():
    """Host side of the values-only path (twr_structure_values_items): every time node of the two grids is in exactly one
    item, an item has at most 64 time nodes and at most eight polynomials of one ee spline in its window, coinciding grids
    fold the range-of-motion rows into the "dynamic" items, a coarse grid is cut at 64 time nodes alone, and structures with
    optimised timings or more than 2046 variables keep the Jacobian kernels' cut."""
    model = ta.model_preset("anymal", "flat")

    def check(S, grids_coincide):
        it = S.values_items()
        assert it["dynamic_takes_rom"] == grids_coincide
        for name, k in (("dynamic", S.k_dynamic), ("rom", 0 if grids_coincide else S.k_rom)):
            nxt = 0
            gather = any(w == 0 for _, _, w in it[name])
            for i, (k0, cnt, widest) in enumerate(it[name]):
                assert k0 == nxt and 1 <= cnt <= 64
                # a coarse grid is cut at 64 time nodes alone (widest 0: its lanes fetch their own records), else by the windows too
                assert (widest == 0 and (cnt == 64 or i == len(it[name]) - 1)) if gather else 1 <= widest <= 8
                nxt = k0 + cnt
            assert nxt == k
        return it

    # BASELINE C3: one dt for both grids, 200 time nodes, 2 s -- four items, none cut by its windows
    dt = 2.0 / (200 - 1.5)
    it = check(ta.Structure(model, ta.gait_combo(4, 1, 2.0), ta.params_default(dt_dynamic=dt, dt_rom=dt)), True)
    assert [c for _, c, _ in it["dynamic"]] == [64, 64, 64, 8] and it["rom"] == []
    # 200 time nodes over 4 s with five polynomials per stance force: the windows cut first (still four items)
    dt = 4.0 / (200 - 1.5)
    it = check(ta.Structure(model, ta.gait_combo(4, 2, 4.0), ta.params_default(dt_dynamic=dt, dt_rom=dt, polys_per_stance_force=5)), True)
    assert any(c < 64 and w == 8 for _, c, w in it["dynamic"][:-1]) and len(it["dynamic"]) == 4
    # towr's default grids (0.1 / 0.08 s): the windows would cut items of a few time nodes -- 64 time nodes each, records per lane
    it = check(ta.Structure(model, ta.gait_combo(4, 1, 2.0), ta.params_default()), False)
    assert it["dynamic"] == [(0, 22, 0)] and it["rom"] == [(0, 27, 0)]
    it = check(ta.Structure(model, ta.gait_combo(4, 2, 8.0), ta.params_default(polys_per_swing=4, polys_per_stance_force=5)), False)
    assert [c for _, c, _ in it["dynamic"]] == [64, 17] and [c for _, c, _ in it["rom"]] == [64, 37]
    # no such items: optimised timings; more than 2046 variables
    for S in (ta.Structure(model, ta.gait_combo(4, 1, 2.0), ta.params_default(constraint_sets=127)),
              ta.Structure(model, ta.gait_combo(4, 1, 24.0), ta.params_default())):
        it = S.values_items()
        assert it["dynamic"] == [] and it["rom"] == [] and not it["dynamic_takes_rom"], S.n
