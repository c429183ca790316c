#!/usr/bin/env python3
"""Benchmark of the hot path: full NLP callbacks (constraint values + Jacobian values) per second.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused HIP kernel over this rank's batch of candidates (inputs already
resident in HBM).  Workload = BASELINE config C3: ANYmal, quadruped combo C1 (flying trot), T=2.0 s,
flat terrain, K_dyn=K_rom=200, `--batch` problems per GPU that share the structure and differ in x
(BASELINE.md section 4 perturbation).  Candidates shard across ranks with no data-path collective
(weak scaling: per-GPU batch fixed); the only collective is one broadcast of the POD model blob.

Prints ONE JSON line on rank 0, with the `roofline` and `cpu_baseline` objects of the contract.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def dist_on(world):
    """torch.distributed is initialised for more than one rank -- or, as a rehearsal of the multi-GPU code path on a
    one-GPU box, when TWR_BENCH_FORCE_DIST is set (world size 1 under torch.distributed.run: the same RCCL init,
    broadcast, barriers, all-gather and all-reduce as on an 8-GPU node)."""
    return world > 1 or bool(os.environ.get("TWR_BENCH_FORCE_DIST"))


LEG_WARMUP_S = 0.25   # device_power_warmup before the untimed warm-up steps of the extra legs (timings_c3, all_sets_c3, scale_c5)


def device_power_warmup(torch, dev, seconds):
    """Brings the GPU out of its idle power state BEFORE the W warm-up steps: plain HBM writes (torch.fill_ of a 1-GiB
    scratch buffer) until `seconds` have passed.  It is not a step of the path and nothing of it is timed.  Why: the set-up
    of a run is seconds of host work during which the device idles, and the clocks need a few hundred milliseconds of load
    to come back -- with W = 5 (7 ms) the timed steps of a fresh process run 4-5 % slower than the same steps a second
    later (A/B on one box, --warmup 3 vs 200: 5.49 / 5.59 vs 5.70 / 5.87 M callbacks/s).  Reported as device_warmup_s."""
    if seconds <= 0:
        return 0.0
    scratch = torch.empty(1 << 27, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            scratch.fill_(1.0)
        torch.cuda.synchronize()
    del scratch
    return time.perf_counter() - t0


from towr_amd.placement import PLACEMENT_BALLAST_GB, place_outputs  # noqa: E402,F401  (a user-side utility: it lives in the package)


def build_case(ta, model, K=200, T=2.0, combo=1, constraint_sets=27):
    sched = ta.gait_combo(model.n_ee, combo, T)
    dt = T / (K - 1.5)  # reference rule floor(T/dt)+2 then yields K nodes
    params = ta.params_default(dt_dynamic=dt, dt_rom=dt, constraint_sets=constraint_sets)
    return sched, params, ta.Structure(model, sched, params)


def perturbed_inputs(S, model, count, first_seed):
    """x0 = reference initial guess (start at nominal stance, goal x=+1 m) + sigma*N(0,1)*scale."""
    ee = [[model.nominal_stance[e][0], model.nominal_stance[e][1], 0.0] for e in range(model.n_ee)]
    z = -model.nominal_stance[0][2]
    x0 = S.initial_guess([0, 0, z], [0, 0, 0], [1.0, 0, z], [0, 0, 0], ee)
    scale = np.ones(S.n)
    for vs in S.var_sets:
        a, b = vs["offset"], vs["offset"] + vs["size"]
        if vs["name"] == "base-lin":
            scale[a:b] = np.tile([0.1] * 3 + [0.5] * 3, vs["size"] // 6)
        elif vs["name"] == "base-ang":
            scale[a:b] = np.tile([0.2] * 3 + [0.5] * 3, vs["size"] // 6)
        elif vs["name"].startswith("ee-motion"):
            scale[a:b] = 0.1
        elif vs["name"].startswith("ee-schedule"):
            scale[a:b] = 0.3  # phase durations: +-15 ms at sigma = 0.05
        else:
            scale[a:b] = 50.0
    out = np.empty((count, S.n))
    for i in range(count):
        rng = np.random.default_rng(1234 + first_seed + i)
        out[i] = x0 + 0.05 * rng.normal(size=S.n) * scale
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Host threads this process may keep busy and where the number comes from: the scheduler affinity mask, cut down by
    a cgroup CPU quota when one is set (v2 cpu.max, v1 cpu.cfs_quota_us); TWR_HOST_CORES overrides both."""
    env = os.environ.get("TWR_HOST_CORES")
    if env:
        return max(1, int(env)), "env"
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    source = "affinity"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = int(f.read()), int(fp.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, source = max(1, int(quota)), "cgroup"
    return max(1, n), source


def ranks_on_node():
    """Ranks that share this node's host cores (torch.distributed.run exports LOCAL_WORLD_SIZE)."""
    return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))


def build_threads():
    """Host threads ONE rank may use for structure builds: this process's share of the node -- the cores it may run on,
    divided by the ranks of the node, at most 64 (eight ranks on a 256-thread node get 32 each, not 64 each)."""
    return max(1, min(64, host_cores()[0] // ranks_on_node()))


def cpu_baseline(sched, params, x, terrain, budget_s=3.0):
    """The oracle ("port" of the reference's Eigen CPU path, reference-shaped: per time node and per variable
    set) timed on this box's host cores, SURVEY 8d / BASELINE.md section 3: built -O3 -march=native ON THIS BOX
    (oracle/Makefile target `native`), (i) one thread, median of 5 samples of >= 2 s each; (ii) one problem
    instance per hardware thread, all running at once (the oracle is not re-entrant, like the reference)."""
    import statistics
    import subprocess
    import threading

    from oracle import binding as ob

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"])
    ob.use_library(os.path.join(ROOT, "oracle", "_build", "libtowr_oracle_native.so"))

    def make():
        return ob.OracleProblem("anymal", terrain, sched.durations(), sched.contact(), dt_dynamic=params.dt_dynamic,
                                dt_rom=params.dt_rom, duration_base_poly=params.duration_base_poly,
                                polys_per_swing=params.polys_per_swing,
                                polys_per_stance_force=params.polys_per_stance_force,
                                constraint_sets=params.constraint_sets)

    P = make()
    t1 = P.time_callbacks(x, 3) / 3.0
    iters = max(5, int(budget_s / max(t1, 1e-6)))
    rates = sorted(iters / P.time_callbacks(x, iters) for _ in range(5))
    single = statistics.median(rates)
    cores, source = host_cores()
    probs = [make() for _ in range(cores)]
    counts = [0] * cores
    chunk = max(1, int(0.25 / max(t1, 1e-6)))   # ~0.25 s of callbacks between looks at the clock
    wall_budget = 4.0
    t0 = time.perf_counter()

    def run(i):   # ctypes drops the GIL for the duration of a call; every instance runs until the common deadline
        while time.perf_counter() - t0 < wall_budget:
            probs[i].time_callbacks(x, chunk)
            counts[i] += chunk

    th = [threading.Thread(target=run, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    all_rate = sum(counts) / wall
    return {"value": single, "unit": "callbacks/s", "cores": 1, "kind": "port",
            "sample": "median of 5 x %d callbacks (%.1f s each) of the same ANYmal K=200 problem, single-thread C++ "
                      "oracle, g++ -O3 -march=native; min %.1f max %.1f" % (iters, iters / single, rates[0], rates[-1]),
            "all_cores": {"value": all_rate, "unit": "callbacks/s", "cores": cores, "cores_source": source,
                          "parallel_efficiency": all_rate / (cores * single),
                          "sample": "%d instances at once for %.1f s, %d callbacks in all (%d .. %d per instance)"
                                    % (cores, wall, sum(counts), min(counts), max(counts))},
            "nproc": os.cpu_count(), "cpu_model": _cpu_model()}


def kernel_source_hash():
    """Identity of the kernels a profile was taken on: sha256 over the device sources and their build flags."""
    import hashlib

    h = hashlib.sha256()
    for name in ("kernels.hip", "rom_tu.hip", "device_tables.h", "Makefile"):
        with open(os.path.join(ROOT, "towr_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def traffic_from_profile(workload, kernel, problems_per_gpu):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc pass (profiles/traffic.json:
    WRITE_SIZE + 2*FETCH_SIZE in KiB, the gfx950 correction of MI355X_MICROARCH.md), or None when the profile
    is for another workload / launch size or was taken on other kernel sources than the ones in this tree."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("workload") == "C3" and t.get("kernel_source_sha256") == kernel_source_hash():
            if workload == "C3" and t.get("problems_per_gpu") == problems_per_gpu:   # (kernel names carry template arguments)
                return sum(v for k, v in t.get("hbm_bytes_per_launch", {}).items() if k.startswith(kernel)) or None
            for wl, key, count in (("C3+timings", "timings_2048", 2048), ("C3+all", "all_sets_8192", 8192)):
                if workload == wl and problems_per_gpu == count:   # the --sets timings --batch 2048 / --sets all passes
                    per = t.get(key, {}).get("hbm_bytes_per_launch", {})
                    if kernel is None:
                        return per
                    return sum(v for k, v in per.items() if k.startswith(kernel)) or None
    except (OSError, ValueError):
        pass
    return None


def traffic_source():
    """Where roofline.traffic comes from: it is NOT measured by this run (PMC counters need rocprofv3 around the process);
    it is the committed profile of the same kernels -- the hash ties it to the kernel sources of this tree."""
    return "profiles/traffic.json@%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, scripts/profile_r05.sh)" % kernel_source_hash()


def sweep_traffic_ratio(algorithmic_bytes):
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        per = t.get("sweep_1024", {}).get("hbm_bytes_per_launch", {})
        if t.get("kernel_source_sha256") == kernel_source_hash() and per:
            # the profiled run holds both forms of a step: the fused launch (what this leg's event-free steps run at 1024
            # candidates) and the three separate kernels (the steps with per-kernel events)
            fused = [v for k, v in per.items() if k.startswith("twr::eval_fused_kernel")]
            return (sum(fused) if fused else sum(per.values())) / algorithmic_bytes
    except (OSError, ValueError):
        pass
    return None


def scale_c5(ta, torch, dist, model, world, rank, dev, dev_index, backend, stream, steps=200, n_total=1024, placement_tries=1):
    """north_star's scaling claim: the 1024-candidate gait / phase-duration sweep on Stairs (BASELINE C5), a FIXED
    total sharded over the ranks (strong scaling).  Every rank builds only its own contiguous shard (cheap
    per-candidate weight, no structure needed to shard), multi-threaded in the library; the steps are timed without
    per-kernel events."""
    from towr_amd import sweep

    m5 = ta.Model.from_buffer_copy(bytes(model))
    m5.terrain_id = ta.TERRAINS["stairs"]
    cands = sweep.enumerate_candidates(n_total)
    threads = build_threads()   # per rank: the node's cores are shared by LOCAL_WORLD_SIZE ranks
    t0 = time.perf_counter()
    # shards by BYTES per callback, exact: every rank computes the same list (pattern only, no device tables)
    weights = sweep.candidate_bytes(m5, cands, threads=threads) if world > 1 else [1.0] * n_total
    bounds = sweep.shard_bounds(weights, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    t_w = time.perf_counter() - t0
    mine = sweep.candidate_structures(m5, cands[lo:hi], threads=threads)
    t1 = time.perf_counter()
    batch = ta.Batch(mine, list(range(len(mine))), device=dev_index)
    setup_s = time.perf_counter() - t0
    x_host = np.concatenate([perturbed_inputs(s_, m5, 1, first_seed=lo + i_)[0] for i_, s_ in enumerate(mine)])
    device_power_warmup(torch, dev, LEG_WARMUP_S)   # (the CPU-baseline leg and the structure builds left the device idle)

    def alloc(jac=None):
        return (torch.from_numpy(x_host).to(dev), torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev),
                jac if jac is not None else torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev))

    def run_steps(bufs, n):
        for _ in range(n):
            batch.eval_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), ta.EVAL_BOTH, stream)

    (x, g, jac), placement = place_outputs(torch, dev, alloc, run_steps, placement_tries, jac_numel=int(batch.jac_off[-1]))   # (as for the headline's buffers)
    for _ in range(10):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    t2 = time.perf_counter()
    for _ in range(steps):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    elapsed = time.perf_counter() - t2
    assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(jac).all())
    stats = torch.tensor([elapsed, setup_s, t1 - t0, float(batch.algorithmic_bytes)], dtype=torch.float64,
                         device=dev if backend == "nccl" else None)
    tot = stats.clone()
    stats0 = stats.clone()
    if dist_on(world):
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed, setup_max, build_max = float(stats[0]), float(stats[1]), float(stats[2])
    bytes_total = float(tot[3])
    # per-rank step time and set-up time (imbalance must be visible, not only the maximum)
    mine_t = torch.tensor([float(stats0[0]) / steps * 1e3, float(stats0[1])], dtype=torch.float64,
                          device=dev if backend == "nccl" else None)
    per_rank = [mine_t.clone() for _ in range(world)]
    if dist_on(world):
        dist.all_gather(per_rank, mine_t)
    per_rank_ms = [float(t_[0]) for t_ in per_rank]
    per_rank_setup = [float(t_[1]) for t_ in per_rank]
    # Planner-style step (optional exchange): constraint values only, scored on the device, one all-gather of
    # the 16 scores per candidate, arg-min ON THE DEVICE (twr_batch_best) -- what a sweep that wants ONE decision does per
    # iterate.  The step is stream-ordered; the decision reaches the host as one 16-byte copy.
    from towr_amd.dist import best_candidate, gather_scores
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    scores = torch.empty((len(mine), 16), dtype=torch.float64, device=dev)
    best_d = torch.zeros(2, dtype=torch.float64, device=dev)
    best_h = torch.zeros(2, dtype=torch.float64).pin_memory()

    def planner_enqueue():   # values -> scores -> this shard's decision (twr_batch_score_best: two launches behind one call)
        batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, stream)
        batch.score_best_device(g.data_ptr(), scores.data_ptr(), best_d.data_ptr(), index_offset=lo, stream=stream)

    def planner_step():   # one decision per step ON THE HOST: the 16-byte copy and its synchronisation are part of the step
        planner_enqueue()
        if dist_on(world):   # every rank's 16-byte decision, all-gathered; the winner on every rank
            from towr_amd.dist import gather_best
            return gather_best(best_d if backend == "nccl" else best_d.cpu()) + (None,)
        best_h.copy_(best_d, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return int(best_h[0]), float(best_h[1]), None

    for _ in range(3):
        best = planner_step()
    # the device's decision is the host rule's (towr_amd.dist.best_candidate) on the all-gathered score table
    torch.cuda.synchronize()
    table = scores
    if dist_on(world):
        table = gather_scores(scores if backend == "nccl" else scores.cpu(), sizes)
    assert (best[0], best[1]) == best_candidate(table), (best[:2], best_candidate(table))
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    p_steps = max(20, steps // 4)
    t3 = time.perf_counter()
    for _ in range(p_steps):
        best = planner_step()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    p_elapsed = torch.tensor([time.perf_counter() - t3], dtype=torch.float64, device=dev if backend == "nccl" else None)
    # ... and back to back, the decision left on the device (a solver loop that consumes it there; one copy at the end)
    t4 = time.perf_counter()
    for _ in range(p_steps):
        planner_enqueue()
    best_h.copy_(best_d, non_blocking=True)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    q_elapsed = torch.tensor([time.perf_counter() - t4], dtype=torch.float64, device=dev if backend == "nccl" else None)
    if dist_on(world):
        dist.all_reduce(p_elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(q_elapsed, op=dist.ReduceOp.MAX)
    p_elapsed, q_elapsed = float(p_elapsed[0]), float(q_elapsed[0])
    curve = shard_curve(ta, torch, sweep, m5, cands, mine, x_host, dev, dev_index, stream, elapsed / steps, threads,
                        placement_tries=min(4, placement_tries)) if world == 1 else None
    return {"workload": "C5 sweep: %d enumerated ANYmal candidates (combo x T x swing scale) on Stairs, K=200, ragged"
                        % n_total,
            "candidates": n_total, "scaling": "strong", "steps": steps,
            "value": n_total * steps / elapsed, "unit": "callbacks/s", "ms_per_step": elapsed / steps * 1e3,
            "path_GBps": bytes_total * steps / elapsed / 1e9,
            "shards": [bounds[r + 1] - bounds[r] for r in range(world)],
            "setup_s": setup_max, "structure_build_s": build_max, "build_threads_per_rank": threads, "ranks_on_node": ranks_on_node(),
            "shard_weights": "8 (n + m + nnz) per candidate (twr_candidate_bytes), %.3f s on this rank" % t_w if world > 1 else "one shard",
            "world_size": world, "backend": backend if dist_on(world) else "none (single process)",
            "rccl_ranks": world if (world > 1 and backend == "nccl") else (1 if world == 1 else 0),
            "device_count": torch.cuda.device_count(),
            "ms_per_step_per_rank": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": per_rank_ms},
            "setup_s_per_rank": per_rank_setup,
            # a shard that re-writes < ~220 MB per step keeps its output in the 256-MB Infinity Cache
            "output_MB_per_rank": float(batch.algorithmic_bytes) / 1e6, "output_placement": placement,
            # HBM traffic of the whole 1024-candidate sweep on ONE GPU over its algorithmic bytes (profiles/traffic.json,
            # rocprofv3 FETCH_SIZE / WRITE_SIZE passes): every candidate reads its own ~150 KB of tables
            "traffic_ratio": sweep_traffic_ratio(bytes_total) if world == 1 else None,
            "traffic_source": traffic_source() if world == 1 else None,
            "planner": {"what": "values only -> twr_batch_score_best (scores, then this shard's arg-min on the device) -> [all-gather of the "
                                "ranks' 16-byte decisions] -> one 16-byte copy to the host + synchronisation per step",
                        "steps": p_steps, "value": n_total * p_steps / p_elapsed, "unit": "candidates scored/s",
                        "ms_per_step": p_elapsed / p_steps * 1e3,
                        "ms_per_step_decision_left_on_device": q_elapsed / p_steps * 1e3, "best_candidate": int(best[0])},
            # what ONE GPU can say about north_star's strong-scaling curve (no collective on the evaluation path)
            "shard_curve": curve}


def shard_curve(ta, torch, sweep, m5, cands, structs, x_host, dev, dev_index, stream, step_s_1024, threads, steps=400, placement_tries=1):
    """The shards an N-rank run of the 1024-candidate sweep cuts (byte-balanced, twr_shard_bounds), each evaluated ON THIS
    GPU, event-free: rank 0's shard and the largest one for N = 2, 4, 8 -- microseconds per step, TB/s, fraction of 8 TB/s
    -- and what the sweep's speed-up would be if nothing but the slowest shard's step mattered.  A PROJECTION from one
    device (no second GPU, no collective, no host contention between ranks), not a scaling measurement."""
    weights = sweep.candidate_bytes(m5, cands, threads=threads)
    x_off = np.concatenate([[0], np.cumsum([s_.n for s_ in structs])])
    out = {"what": "byte-balanced shards of the 1024 candidates for world = 2, 4, 8, each timed on this one GPU (event-free "
                   "steps); projected_speedup_no_overhead = step(1024) / step(slowest of the shards timed) -- a projection, "
                   "not a scaling measurement", "steps": steps, "step_us_1024": step_s_1024 * 1e6, "world": {}}
    for world in (2, 4, 8):
        bounds = sweep.shard_bounds(weights, world)
        sizes = [sum(weights[bounds[r]:bounds[r + 1]]) for r in range(world)]
        picks = sorted({0, int(np.argmax(sizes))})
        rows = []
        for r in picks:
            lo, hi = bounds[r], bounds[r + 1]
            batch = ta.Batch(structs[lo:hi], list(range(hi - lo)), device=dev_index)
            xh = x_host[x_off[lo]:x_off[hi]].copy()

            def alloc(jac=None):
                return (torch.from_numpy(xh).to(dev), torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev),
                        jac if jac is not None else torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev))

            def run_steps(bufs, n):
                for _ in range(n):
                    batch.eval_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), ta.EVAL_BOTH, stream)

            (x, g, jac), placed = place_outputs(torch, dev, alloc, run_steps, placement_tries, jac_numel=int(batch.jac_off[-1]))   # (as for the 1024-candidate step)
            for _ in range(20):
                batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(jac).all())
            rows.append({"rank": r, "candidates": hi - lo, "bytes": int(batch.algorithmic_bytes), "us_per_step": dt * 1e6,
                         "TBps": batch.algorithmic_bytes / dt / 1e12, "frac": batch.algorithmic_bytes / dt / 1e9 / HBM_PEAK_GBS,
                         "largest": r == int(np.argmax(sizes)),
                         "placement_tries_us": [round(t_["ms_per_step"] * 1e3, 1) for t_ in placed["tries"] if t_.get("ms_per_step") is not None]})
            del batch, x, g, jac
        slowest = max(r_["us_per_step"] for r_ in rows)
        out["world"][str(world)] = {"shards": rows, "projected_speedup_no_overhead": step_s_1024 * 1e6 / slowest}
    return out


def values_c3(ta, torch, model, dev, dev_index, stream, steps=50, B=8192):
    """What Ipopt calls most (every line-search trial point is an eval_g): the C3 callback with TWR_EVAL_VALUES at batch
    size, per-kernel events.  Algorithmic bytes = 8 (n + m) per problem: x read once, g written once -- 36 KB against the
    859 KB of the full callback, so HBM is not the bound; the leg reports the fraction of both roofs."""
    m = ta.Model.from_buffer_copy(bytes(model))
    m.terrain_id = ta.TERRAINS["flat"]
    sched, params, S = build_case(ta, m)
    batch = ta.Batch([S], [0] * B, device=dev_index)
    base = perturbed_inputs(S, m, min(B, 256), first_seed=0)
    x = torch.from_numpy(np.tile(base, ((B + base.shape[0] - 1) // base.shape[0], 1))[:B].reshape(-1)).to(dev)
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev)
    device_power_warmup(torch, dev, LEG_WARMUP_S)
    for _ in range(5):
        batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, stream)
    batch.profile_begin(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, n_prof = batch.profile_end()
    assert n_prof == steps and bool(torch.isfinite(g).all())
    # event-free (what a solver loop sees)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.eval_device(x.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, stream)
    torch.cuda.synchronize()
    free_ms = (time.perf_counter() - t0) / steps * 1e3
    # (with per-kernel events the lane-per-node items of both sets are still ONE launch: the second interval is empty, ~5 us of
    # event overhead; the node sets follow in a launch of their own)
    names = {"dynamic": "twr::eval_values_kernel (dynamic + rangeofmotion items)", "rangeofmotion": "(empty interval)", "nodes": "twr::node_kernel2 (values)"}
    if kern_ms.get("rangeofmotion", 1.0) < 0.02:
        kern_ms.pop("rangeofmotion")
    bytes_values = 8 * (S.n + S.m) * B
    path_ms = sum(kern_ms.values())
    # FP64 vector peak: 256 CUs x 4 SIMDs x 16 FP64 lanes/clk x 2 (FMA) x 2.4 GHz = 78.6 TFLOP/s = half the guide's FP32
    # vector rate (157.3 TF); VALU instructions per step from the committed PMC pass (profiles/traffic.json,
    # SQ_INSTS_VALU of the event-free launch, eval_values_kernel: dynamic + range of motion + node sets), each charged as a
    # 4-cycle wave64 FP64 issue -- an UPPER bound of the pipe's occupancy
    valu = valu_from_profile(B)
    issue_peak = 256 * 4 * 2.4e9 / 4.0   # wave64 FP64 instructions per second, chip-wide
    valu_frac = valu / (free_ms * 1e-3) / issue_peak if valu else None
    return {"workload": "C3 values only (TWR_EVAL_VALUES): n=%d m=%d, %d problems/GPU" % (S.n, S.m, B), "problems_per_gpu": B,
            "steps": steps, "value": B * steps / elapsed, "unit": "callbacks/s", "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_without_events": free_ms, "bytes_per_callback": 8 * (S.n + S.m),
            "kernel_ms": {names[k]: v for k, v in kern_ms.items()},
            "hbm": {"achieved": bytes_values / (path_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_values / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "frac_without_events": bytes_values / (free_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "fp64_valu": {"peak_TFLOPs": 78.6, "issue_frac_upper_bound": valu_frac, "valu_insts_per_step": valu,
                          "source": "SQ_INSTS_VALU of profiles/traffic.json values_8192 (rocprofv3 --pmc; the one-launch step), 4 cycles per "
                                    "wave64 instruction, over ms_per_step_without_events" if valu_frac else None},
            "bound": (("vector-instruction issue" if valu_frac >= 0.6 else "neither roof alone (per-wave latency: a wave is parked at a wait or the group "
                       "barrier for more than half of its life, profiles/r05_pmc_summary.txt)")
                      + ": %.0f %% of the issue slots at 2.4 GHz (every wave64 FP64 instruction holds its SIMD for four cycles) against %.0f %% of HBM"
                      % (100 * valu_frac, 100 * bytes_values / (free_ms * 1e-3) / 1e9 / HBM_PEAK_GBS))
                     if valu_frac else "HBM fraction above; no VALU instruction counts for these kernel sources (profiles/traffic.json is of another tree)"}


def valu_from_profile(problems_per_gpu):
    """VALU wave-instructions of one event-free values-only step (the single eval_values_kernel launch)
    from the committed PMC pass, or None when the profile is of other kernel sources / another batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        v = t.get("values_8192", {})
        if t.get("kernel_source_sha256") == kernel_source_hash() and v.get("problems_per_gpu") == problems_per_gpu:
            return v.get("valu_insts_per_step") or None
    except (OSError, ValueError):
        pass
    return None


def timings_c3(ta, torch, dist, model, world, rank, dev, dev_index, backend, stream, steps=20, B=2048, constraint_sets=127, placement_tries=1):
    """Another constraint list on the same C3 problem, driver-timed beside the headline: constraint_sets = 127 is the
    optimised-timings mode (Parameters::OptimizePhaseDurations: ee-schedule variables, x-dependent active polynomials,
    rows that hold every variable of every ee set; 2048 problems per GPU), 63 is towr's whole default
    constraint list (+ splineacc-base-*, swing-*; 8192 problems per GPU).  Per-kernel HIP events, roofline of the dominant
    kernel and of the whole path."""
    m = ta.Model.from_buffer_copy(bytes(model))
    m.terrain_id = ta.TERRAINS["flat"]
    sched, params, S = build_case(ta, m, constraint_sets=constraint_sets)
    batch = ta.Batch([S], [0] * B, device=dev_index)
    base = perturbed_inputs(S, m, min(B, 256), first_seed=rank * 100000)
    x_host = np.tile(base, ((B + base.shape[0] - 1) // base.shape[0], 1))[:B].reshape(-1)
    device_power_warmup(torch, dev, LEG_WARMUP_S)

    def alloc(jac=None):
        return (torch.from_numpy(x_host).to(dev), torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev),
                jac if jac is not None else torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev))

    def run_steps(bufs, n):
        for _ in range(n):
            batch.eval_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), ta.EVAL_BOTH, stream)

    (x, g, jac), placement = place_outputs(torch, dev, alloc, run_steps, placement_tries, jac_numel=int(batch.jac_off[-1]))   # (as for the headline's buffers)
    for _ in range(3):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)
    batch.profile_begin(steps)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, n_prof = batch.profile_end()
    assert n_prof == steps
    assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(jac).all())
    if dist_on(world):
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timings = bool(constraint_sets & 64)
    # (batches of >= 2048 problems with splineacc / swing rows and fixed timings: the persistent node_chunk_kernel)
    names = {"dynamic": "twr::phase_locate_kernel + twr::dyn_phase_kernel" if timings else "twr::dyn_kernel",
             "rangeofmotion": "twr::rom_phase_kernel" if timings else "twr::rom_kernel",
             "nodes": "twr::node_kernel" if (timings or B < 2048) else "twr::node_chunk_kernel"}   # (8 problems per CU)
    kbytes = batch.kernel_bytes()
    dom = max(kern_ms, key=kern_ms.get)
    path_ms = sum(kern_ms.values())
    traffic = traffic_from_profile("C3+timings" if timings else "C3+all", None, B)
    dom_traffic = None
    if traffic:   # the dynamic interval holds the pre-pass and the kernel: both are charged
        keys = {"dynamic": ["twr::dyn_phase_kernel", "twr::phase_locate_kernel"] if timings else ["twr::dyn_kernel"],
                "rangeofmotion": ["twr::rom_phase_kernel" if timings else "twr::rom_kernel"], "nodes": ["twr::node_kernel"]}[dom]
        vals = [v for k, v in traffic.items() if any(k.startswith(q) for q in keys)]
        dom_traffic = sum(vals) if vals else None
    return {"workload": "C3 with %s: n=%d m=%d nnz=%d, %d problems/GPU"
                        % ("optimised phase durations" if timings else "towr's whole default constraint list", S.n, S.m, S.nnz, B),
            "problems_per_gpu": B, "steps": steps, "value": B * world * steps / elapsed, "unit": "callbacks/s",
            "ms_per_step": elapsed / steps * 1e3, "bytes_per_callback": S.algorithmic_bytes, "output_placement": placement,
            "roofline": {"bound": "hbm", "achieved": kbytes[dom] / (kern_ms[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kbytes[dom] / (kern_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": dom_traffic,
                         "traffic_source": traffic_source() if dom_traffic else None,
                         "kernel": names[dom], "kernel_ms": kern_ms[dom], "algorithmic_bytes_per_launch": kbytes[dom],
                         "path": {"achieved": batch.algorithmic_bytes / (path_ms * 1e-3) / 1e9,
                                  "frac": batch.algorithmic_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "kernel_ms": {names[k]: v for k, v in kern_ms.items()},
                                  "algorithmic_bytes_per_step": batch.algorithmic_bytes}}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="candidates per GPU (default 8192, sweep: 128)")
    ap.add_argument("--workload", choices=["c3", "sweep"], default="c3",
                    help="c3: BASELINE config 3 (one schedule, distinct x, the headline metric); sweep: BASELINE "
                         "configs 4-5 (enumerated gait / duration candidates on Stairs, ragged structures)")
    ap.add_argument("--sets", choices=["hot", "all", "timings"], default="hot",
                    help="hot: the four constraint families of the headline metric (BASELINE sizes n=640 m=3866 "
                         "nnz=102896); all: towr's whole default list (+ splineacc-base-*, swing-*); timings: all + "
                         "optimised phase durations (ee-schedule variables, all-variables rows); c3 only")
    ap.add_argument("--device-warmup-s", type=float, default=0.5,
                    help="seconds of plain HBM writes before the W warm-up steps, to leave the idle power state (0: none)")
    ap.add_argument("--placement-tries", type=int, default=12,
                    help="allocate the output buffers this many times at different places of device memory and keep the fastest "
                         "(untimed probe steps, before the W warm-up steps; 1: as the allocator hands them out)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale-c5", action="store_true", help="skip the strong-scaling C5 leg of the default run")
    ap.add_argument("--no-values-c3", action="store_true", help="skip the values-only leg of the default run")
    ap.add_argument("--no-timings-c3", action="store_true",
                    help="skip the optimised-timings and the whole-default-list legs of the default run")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import towr_amd as ta

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # rehearsal knobs (one-GPU box): TWR_BENCH_BACKEND=gloo TWR_BENCH_DEVICE=0 run every rank on one device
    backend = os.environ.get("TWR_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("TWR_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    # --- the single collective of the design: rank 0 broadcasts the POD robot/terrain model (RCCL)
    if dist_on(world):
        from towr_amd.dist import broadcast_model

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        model = broadcast_model(ta.model_preset("anymal", "flat") if rank == 0 else None, src=0,
                                device=dev if backend == "nccl" else None)
    else:
        model = ta.model_preset("anymal", "flat")

    terrain = "flat" if args.workload == "c3" else "stairs"
    model.terrain_id = ta.TERRAINS[terrain]  # (the broadcast blob carries the robot; the workload picks the terrain)
    t_setup = time.perf_counter()
    if args.workload == "c3":
        sched, params, S = build_case(ta, model, constraint_sets={"hot": 27, "all": 63, "timings": 127}[args.sets])
        B = args.batch or 8192
        n_all = B * world
        workload = "C3 ANYmal trot (combo C1), T=2.0 s, flat, K=%d, n=%d m=%d nnz=%d, %d problems/GPU (256 seeded x tiled)%s" % (
            S.k_dynamic, S.n, S.m, S.nnz, B, {"hot": "", "all": ", all six default constraint families",
                                              "timings": ", default constraints + optimised phase durations"}[args.sets])
        batch = ta.Batch([S], [0] * B, device=dev_index)
        setup_s = time.perf_counter() - t_setup
        # distinct x per problem: 256 seeded perturbations per rank, tiled (contents do not change the work)
        base = perturbed_inputs(S, model, min(B, 256), first_seed=rank * 100000)
        reps = (B + base.shape[0] - 1) // base.shape[0]
        x_host = np.tile(base, (reps, 1))[:B].reshape(-1)
        bytes_per_callback = S.algorithmic_bytes
    else:
        # C4/C5: the canonical candidate list, contiguous shards balanced by bytes; every candidate has its
        # own structure (SURVEY 8d enumeration, towr_amd/sweep.py)
        from towr_amd import sweep

        n_all = (args.batch or 128) * world
        cands = sweep.enumerate_candidates(n_all)
        t_setup = time.perf_counter()
        bounds = sweep.shard_bounds(sweep.candidate_bytes(model, cands, threads=build_threads()) if world > 1 else [1.0] * n_all, world)
        mine = sweep.candidate_structures(model, cands[bounds[rank]:bounds[rank + 1]], threads=build_threads())   # this rank's shard only
        batch = ta.Batch(mine, list(range(len(mine))), device=dev_index)
        setup_s = time.perf_counter() - t_setup
        x_host = np.concatenate([perturbed_inputs(s_, model, 1, first_seed=bounds[rank] + i_)[0]
                                 for i_, s_ in enumerate(mine)])
        S, sched, params = mine[0], mine[0].schedule, mine[0].params
        B = len(mine)
        workload = "C5 sweep: %d enumerated ANYmal candidates (combo x T x swing scale) on Stairs, K=200, ragged" % n_all
        bytes_per_callback = batch.algorithmic_bytes // B
    stream = torch.cuda.current_stream().cuda_stream
    global LEG_WARMUP_S
    LEG_WARMUP_S = min(LEG_WARMUP_S, args.device_warmup_s)   # (--device-warmup-s 0 switches every such warm-up off)
    warm_s = device_power_warmup(torch, dev, args.device_warmup_s)

    def alloc(jac=None):
        return (torch.from_numpy(x_host).to(dev), torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device=dev),
                jac if jac is not None else torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device=dev))

    def run_steps(bufs, n):
        for _ in range(n):
            batch.eval_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), ta.EVAL_BOTH, stream)

    (x, g, jac), placement = place_outputs(torch, dev, alloc, run_steps, args.placement_tries, jac_numel=int(batch.jac_off[-1]))

    def step():
        batch.eval_device(x.data_ptr(), g.data_ptr(), jac.data_ptr(), ta.EVAL_BOTH, stream)

    for _ in range(args.warmup):
        step()
    # HIP events on the launch stream bracket each of the three kernels of every timed step
    batch.profile_begin(args.steps)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, n_prof = batch.profile_end()
    assert n_prof == args.steps
    # measurement hygiene: the same timed region twice more (same K steps, same per-kernel events, barriers on both
    # sides).  The headline stays the FIRST region (the contract's K steps); the repeats show its run-to-run spread.
    repeats = [elapsed]
    for _ in range(2):
        batch.profile_begin(args.steps)
        torch.cuda.synchronize()
        if dist_on(world):
            dist.barrier()
        t_r = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist_on(world):
            dist.barrier()
        repeats.append(time.perf_counter() - t_r)
        batch.profile_end()
    if dist_on(world):
        t_rep = torch.tensor(repeats, dtype=torch.float64, device=dev if backend == "nccl" else None)
        dist.all_reduce(t_rep, op=dist.ReduceOp.MAX)
        repeats = [float(v) for v in t_rep]

    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist_on(world):
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else None)
        every = [t.clone() for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(e_[0]) / args.steps * 1e3 for e_ in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the outputs are finite (no work skipped / no garbage)
    assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(jac).all())

    if rank == 0:
        callbacks = n_all * args.steps
        alg_bytes = batch.algorithmic_bytes  # 8*(n+m+nnz) per problem x problems per step
        kbytes = batch.kernel_bytes()
        dom = max(kern_ms, key=kern_ms.get)  # dominant kernel by measured time
        achieved = kbytes[dom] / (kern_ms[dom] * 1e-3) / 1e9
        path_ms = sum(kern_ms.values())
        path_achieved = alg_bytes / (path_ms * 1e-3) / 1e9
        names = {"dynamic": "twr::dyn_kernel", "rangeofmotion": "twr::rom_kernel",
                 # batches that carry the hot-path sets only run the two-family node kernel
                 "nodes": "twr::node_kernel2" if (args.workload != "c3" or args.sets == "hot") else
                          ("twr::node_chunk_kernel" if (args.sets == "all" and B >= 2048) else "twr::node_kernel")}
        if args.workload == "c3" and args.sets == "timings":
            names.update(dynamic="twr::dyn_phase_kernel", rangeofmotion="twr::rom_phase_kernel")
        main_traffic = (traffic_from_profile({"hot": "C3", "timings": "C3+timings", "all": "C3+all"}[args.sets], names[dom], B)
                        if args.workload == "c3" else None)
        out = {
            "metric": "constraint+Jacobian evals/sec (full NLP callback), 4-EE SRBD",
            "value": callbacks / elapsed,
            "unit": "callbacks/s",
            "n_gpus": world,
            "backend": backend if dist_on(world) else "none (single process)", "device_count": torch.cuda.device_count(),
            "ms_per_step_per_rank": per_rank_ms,
            "steps": args.steps,
            "warmup": args.warmup,
            "device_warmup_s": warm_s,   # plain HBM writes before the W warm-up steps (idle power state), not steps of the path
            "ms_per_step": elapsed / args.steps * 1e3,
            # the timed region three times in this process (max over ranks each); [0] is the headline's region
            "ms_per_step_repeats": [r_ / args.steps * 1e3 for r_ in repeats],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "problems_per_gpu": B, "bytes_per_callback": bytes_per_callback,
                       "setup_s": setup_s, "output_placement": placement},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": main_traffic,
                         "kernel": names[dom], "kernel_ms": kern_ms[dom],
                         "algorithmic_bytes_per_launch": kbytes[dom],
                         "traffic_source": traffic_source() if main_traffic else None,
                         # the whole callback = the three kernels back to back (SURVEY 8d figure 8*(n+m+nnz))
                         "path": {"achieved": path_achieved, "frac": path_achieved / HBM_PEAK_GBS,
                                  "kernel_ms": {names[k]: v for k, v in kern_ms.items()},
                                  "algorithmic_bytes_per_step": alg_bytes}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sched, params, x_host[:S.n], terrain)
    # north_star's strong-scaling claim rides on the same command: 1024 sweep candidates over all ranks
    c5 = t3 = None
    default_run = args.workload == "c3" and args.sets == "hot"
    if default_run and not (args.no_scale_c5 and args.no_timings_c3 and args.no_values_c3):
        del x, g, jac, batch
    a3 = v3 = None
    if default_run and not args.no_values_c3 and world == 1:
        v3 = values_c3(ta, torch, model, dev, dev_index, stream)
    if default_run and not args.no_timings_c3:
        tries = min(4, args.placement_tries)
        t3 = timings_c3(ta, torch, dist, model, world, rank, dev, dev_index, backend, stream, placement_tries=tries)
        a3 = timings_c3(ta, torch, dist, model, world, rank, dev, dev_index, backend, stream, B=8192, constraint_sets=63, placement_tries=tries)
    if default_run and not args.no_scale_c5:
        c5 = scale_c5(ta, torch, dist, model, world, rank, dev, dev_index, backend, stream, placement_tries=min(8, args.placement_tries))
    if rank == 0:
        if t3 is not None:
            out["timings_c3"] = t3
            out["all_sets_c3"] = a3
        if v3 is not None:
            out["values_c3"] = v3
        if c5 is not None:
            out["scale_c5"] = c5
        print(json.dumps(out), flush=True)
    if dist_on(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
