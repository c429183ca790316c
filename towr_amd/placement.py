"""Where the output buffers of a batch sit in device memory (a caller-side utility; bench.py uses it for its own buffers).

The library evaluates into buffers the CALLER allocates.  On MI355X the same binary, batch and step count run 5-10 % apart depending
on nothing but which PHYSICAL device memory a buffer happened to get: the device's memory has faster and
slower stretches for many concurrent store streams, each some tens of GB long -- a store-only micro-benchmark on 32 buffers of
6.7 GB held together writes seven of them at 6.0-6.4 TB/s and twenty-five at 5.6-5.8 TB/s, the same ones in every pass
(scripts/probe/wr_alloc_probe.hip, profiles/r05_write_rate_by_allocation.txt), and inside ONE 200-GB allocation the C3 step takes
1.38-1.40 ms in some windows and 1.50-1.58 ms in others (scripts/arena_probe.py).  The state belongs to the memory behind the
mapping: time under load does not change it, the way the memory is mapped does not matter, the same virtual address is fast in one
allocation and slow in the next; address translation, store policy and residency have nothing to do with it.  A caller that
evaluates into the same buffers millions of times can afford to look once: place_outputs(..., jac_numel=n) makes the Jacobian buffer
a window of one large arena and keeps the fastest of a few dozen windows (place_in_arena); without jac_numel it allocates the
buffers several times, each behind a ballast that is freed again, and keeps the fastest allocation."""
import time

PLACEMENT_BALLAST_GB = (0.0, 2.0, 5.0, 10.0, 1.0, 3.0, 7.0, 14.0)   # what is held while the buffers of placement try i are allocated


_ARENAS = {}   # device -> the arena tensor of this process (kept until the process ends)


def _arena(torch, dev, fraction=0.7):
    """One large allocation per device and process -- `fraction` of what is free at the first call -- whose windows serve as Jacobian
    buffers (place_outputs with jac_numel).  None when it cannot be had."""
    key = str(dev)
    if key not in _ARENAS:
        try:
            free, _total = torch.cuda.mem_get_info(dev)
            _ARENAS[key] = torch.empty(int(fraction * free) // (1 << 21) * (1 << 18), dtype=torch.float64, device=dev)
        except Exception:   # noqa: BLE001  (no mem_get_info, or the allocation failed: the caller falls back)
            _ARENAS[key] = None
    return _ARENAS[key]


def place_in_arena(torch, dev, alloc, run_steps, tries, jac_numel):
    """The Jacobian buffer as a WINDOW of one large arena allocation: 2 x `tries` windows spread evenly over the arena, a few untimed
    steps into each, the fastest kept (inside a 200-GB allocation the C3 step takes 1.38-1.40 ms in windows
    that lie in the faster stretches of the device's memory and 1.50-1.58 ms in the slower ones, each some tens of GB long --
    scanning one arena finds a fast stretch on every box, which allocating a few candidates does not).  alloc(jac) makes the
    (x, g, jac) triple around a given Jacobian tensor.  Returns None when there is no arena (the caller then places by allocation)."""
    arena = _arena(torch, dev)
    if arena is None or arena.numel() < jac_numel:
        return None
    k = max(2, 2 * tries)
    span = arena.numel() - jac_numel
    offsets = sorted({int(span * i / (k - 1)) // 32 * 32 for i in range(k)})
    first, best, report = None, None, []
    for off in offsets:
        jac = arena[off:off + jac_numel]
        if first is None:
            first = alloc(jac)
        bufs = (first[0], first[1], jac)
        for _ in range(3):
            run_steps(bufs, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(bufs, 10)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        report.append({"offset_GB": round(off * 8 / 2 ** 30, 1), "ms_per_step": ms})
        if best is None or ms < best[0]:
            best = (ms, bufs, len(report) - 1)
    return best[1], {"tries": report, "kept": best[2], "arena_GB": round(arena.numel() * 8 / 2 ** 30, 1),
                     "what": "the Jacobian buffer is a window of ONE arena allocation of this process (0.7 of the free device memory): "
                             "2 x `tries` windows spread over it, ten untimed steps into each, the fastest kept "
                             "(towr_amd.placement.place_in_arena: the device's memory has faster and slower stretches for the kernels' "
                             "store streams, 5-10 % apart, DESIGN 6.R5); --placement-tries 1 switches it off"}


def place_outputs(torch, dev, alloc, run_steps, tries, jac_numel=None):
    """With jac_numel (and tries > 1): place_in_arena -- alloc then takes the Jacobian tensor to build the triple around, alloc(jac);
    without an arena, or without jac_numel, placement by allocation, as follows (alloc() / alloc(None) allocates all three).
    Some large output allocations of a process are 5-10 % slower to evaluate into than any later allocation
    (profiles/r05_output_placement_probe.txt): the same binary, batch and addresses, `rom_kernel` 0.94 ms on them
    and 0.85 ms on buffers allocated after them -- whatever is held in between (ballast of 0 ... 112 GB), while a plain
    torch.fill_ of the same buffer runs at 6.75 TB/s either way.  Keeping the first buffers and running for seconds does not
    help, re-allocating them with nothing else in between does not either; allocating, freeing and allocating again behind
    another allocation often does.  The state belongs to the allocation (every offset inside a slow one is slow; backing it with
    2-MiB or 1-GiB physical chunks through the HIP virtual-memory API changes nothing); why is not established.  It is what
    earlier rounds had filed as "the box" (5.3 vs 5.7 M callbacks/s).
    A caller that evaluates into the same buffers millions of times allocates them once and can afford to look: this
    allocates the buffers `tries` times -- each time behind a ballast allocation of another size, which is freed again --,
    runs a few untimed steps on each and keeps the fastest.  Nothing of it is inside the timed region; the line reports
    every try.  tries = 1: the buffers as the allocator hands them out."""
    if jac_numel is not None and tries > 1:
        placed = place_in_arena(torch, dev, alloc, run_steps, tries, jac_numel)
        if placed is not None:
            return placed
    best, report = None, []
    for i in range(max(1, tries)):
        if i > 0:   # a second placement must fit beside the one that is kept, plus its ballast: otherwise stop trying
            free, _total = torch.cuda.mem_get_info(dev)
            next_gb = PLACEMENT_BALLAST_GB[i % len(PLACEMENT_BALLAST_GB)] + 20.0 * (i // len(PLACEMENT_BALLAST_GB))
            if free < 1.25 * kept_bytes + next_gb * 2 ** 30:
                report.append({"ballast_GB": next_gb, "ms_per_step": None, "skipped": "not enough free device memory for another placement"})
                break
        gb = PLACEMENT_BALLAST_GB[i % len(PLACEMENT_BALLAST_GB)] + 20.0 * (i // len(PLACEMENT_BALLAST_GB))
        ballast = torch.empty(int(gb * (1 << 27)), dtype=torch.float64, device=dev) if gb > 0 else None
        before = torch.cuda.mem_get_info(dev)[0] if hasattr(torch.cuda, "mem_get_info") else 0
        bufs = alloc()
        kept_bytes = max(0, before - (torch.cuda.mem_get_info(dev)[0] if hasattr(torch.cuda, "mem_get_info") else 0))
        del ballast
        if tries > 1:
            for _ in range(3):
                run_steps(bufs, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(bufs, 10)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
        else:
            ms = None
        report.append({"ballast_GB": gb, "ms_per_step": ms})
        if best is None or (ms is not None and ms < best[0]):
            best = (ms, bufs, i)
        del bufs
        torch.cuda.empty_cache()
    return best[1], {"tries": report, "kept": best[2],
                     "what": "x / g / Jacobian buffers allocated `tries` times (each behind a ballast allocation that is freed again), ten "
                             "untimed steps on each, the fastest kept (towr_amd.placement.place_outputs: the first large allocation of a process "
                             "evaluates 5-10 % slower than later ones); --placement-tries 1 switches it off"}
