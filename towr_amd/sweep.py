"""Candidate contact-sequence sweeps (BASELINE configs C4/C5) and their sharding over ranks.

The reference has no sweep code (fpowr solves one hard-coded gait per goal,
fpowr/src/footstep_plan_server.cc:191-200); the candidates are enumerated deterministically from the
reference's GaitGenerator tables as SURVEY.md section 8(d) prescribes:
lexicographic over combo C0..C4 x T_i = 1.2 + 0.2 i (i = 0..7) x swing scale s_j = 0.80 + 0.016 j
(j = 0..25), truncated to the first B.  Every candidate has its own structure (ragged n/m/nnz).
"""
import numpy as np

from . import Structure, gait_combo, params_default


def enumerate_candidates(count, n_ee=4):
    """[(combo, T, swing_scale)] in the canonical order; 1040 candidates exist."""
    out = []
    for combo in range(5):
        for i in range(8):
            for j in range(26):
                out.append((combo, 1.2 + 0.2 * i, 0.80 + 0.016 * j))
    if count > len(out):
        raise ValueError("only %d candidates are defined" % len(out))
    return out[:count]


def candidate_inputs(model, cand, k_nodes=200, constraint_sets=None):
    """(schedule, params) of one candidate: GaitGenerator tables scaled to T, K time nodes."""
    combo, T, scale = cand
    sched = gait_combo(model.n_ee, combo, T, scale)
    dt = T / (k_nodes - 1.5)  # the reference rule floor(T/dt)+2 then gives k_nodes time nodes
    kw = {} if constraint_sets is None else dict(constraint_sets=constraint_sets)
    return sched, params_default(dt_dynamic=dt, dt_rom=dt, **kw)


def candidate_structure(model, cand, k_nodes=200, grid=None, constraint_sets=None):
    return Structure(model, *candidate_inputs(model, cand, k_nodes, constraint_sets), grid=grid)


def candidate_structures(model, cands, k_nodes=200, threads=0, grid=None, constraint_sets=None):
    """The structures of a list of candidates, built in one multi-threaded library call
    (twr_structure_create_many[_with_grid]; `grid`: the shared gridded terrain of the sweep).  A rank calls this for
    ITS shard only."""
    inputs = [candidate_inputs(model, c, k_nodes, constraint_sets) for c in cands]
    return Structure.create_many(model, [i[0] for i in inputs], [i[1] for i in inputs], threads, grid=grid)


def candidate_weight(cand, k_nodes=200):
    """Cheap stand-in weight of ONE candidate without building anything: the time-node count (the value array is
    dominated by K time nodes x (dynamic + 4 rangeofmotion rows)).  Good enough when every candidate has the same K and
    constraint sets (nnz then varies by < 0.2 % over the enumeration); sweeps that mix either use candidate_bytes."""
    return float(k_nodes)


def candidate_bytes(model, cands, k_nodes=200, threads=0, constraint_sets=None):
    """Exact bytes per callback, 8 (n + m + nnz), of every candidate -- the weight SURVEY 8e shards by -- from the
    library (twr_candidate_bytes: variable layout + time tables + CSR pattern only, multi-threaded, no device tables).
    `k_nodes` may be one number or one per candidate.  Every rank computes the same list."""
    import ctypes as C

    from . import Params, Schedule, _check, lib

    ks = [k_nodes] * len(cands) if np.isscalar(k_nodes) else list(k_nodes)
    inputs = [candidate_inputs(model, c, k, constraint_sets) for c, k in zip(cands, ks)]
    n = len(inputs)
    sa = (Schedule * n)(*[i[0] for i in inputs])
    pa = (Params * n)(*[i[1] for i in inputs])
    out = np.zeros(n, dtype=np.int64)
    _check(lib().twr_candidate_bytes(C.byref(model), sa, pa, n, int(threads), out.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


def shard_bounds(weights, world):
    """Contiguous shards balanced by the prefix sum of `weights` (bytes per candidate):
    rank r owns [bounds[r], bounds[r+1]).  Every candidate lands in exactly one shard, no shard is empty.
    (twr_shard_bounds of the C ABI: the C++ caller, examples/sweep_multi_gpu.cc, cuts the same shards.)"""
    import ctypes as C

    from . import TowrError, lib

    w = np.ascontiguousarray(weights, dtype=np.float64)
    if world < 1 or world > len(w):
        # every rank evaluates this with the same arguments, so all of them raise (no rank is left in a collective)
        raise ValueError("cannot shard %d candidates over %d ranks: every rank needs at least one" % (len(w), world))
    out = np.zeros(world + 1, dtype=np.int32)
    rc = lib().twr_shard_bounds(w.ctypes.data_as(C.POINTER(C.c_double)), len(w), int(world), out.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise TowrError(lib().twr_last_error().decode())
    return [int(b) for b in out]
