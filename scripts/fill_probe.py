#!/usr/bin/env python3
"""HBM write ceilings on this box for comparison with the kernels' store streams: torch.fill_ (one contiguous
grid-stride stream) over buffers of different sizes -- the rate falls with the footprint."""
import sys, time
import torch
sizes = [float(s) for s in sys.argv[1:]] or [0.5, 1.0, 2.35, 3.5, 4.63, 7.0]
big = torch.empty(int(max(sizes) * 1e9 / 8), dtype=torch.float64, device="cuda")
for gb in sizes:
    a = big[: int(gb * 1e9 / 8)]
    for _ in range(3):
        a.fill_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        a.fill_(2.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("fill_ %.2f GB: %.3f ms = %.2f TB/s" % (gb, dt * 1e3, a.numel() * 8 / dt / 1e12), flush=True)
