#!/usr/bin/env python3
"""Runtime ablations of the dyn kernel on the diagnostic build (make -C towr_amd/csrc ablate): which part of a slice's
life costs what.  Flags (TWR_DEBUG_FLAGS): 0x100 no copy-out, 0x200 no back half, 0x400 no front half (no xs reads),
0x1000 back half computes but stores nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "towr_amd", os.environ.get("ABL_LIB", "libtowr_amd_ablate.so"))
cases = [("full", 0), ("no copy-out", 0x100), ("no back", 0x200), ("no front", 0x400), ("no front/back", 0x600),
         ("back without LDS stores", 0x1000), ("nothing", 0x700), ("no x loads (rom)", 0x400 | 0x2000),
         ("clamped overshoot stores (old)", 0x8000), ("copy-out only, old clamp", 0x8600), ("copy-out only, const data", 0x4600),
         ("no copy-out, no front", 0x500), ("no copy-out, no back", 0x300), ("no copy-out, back w/o LDS stores", 0x1100)]
if os.environ.get("ABL_CASES"):
    want = os.environ["ABL_CASES"].split(";")
    cases = [c for c in cases if c[0] in want]
extra = sys.argv[1:]
for name, fl in cases:
    env = dict(os.environ, TWR_AMD_LIB=lib, TWR_DEBUG_FLAGS=str(fl))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline",
                        "--no-scale-c5"] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        k = d["roofline"]["path"]["kernel_ms"]
        print("%-22s flags %#6x  dyn %.3f ms  rom %.3f ms" % (name, fl, k["twr::dyn_kernel"], k["twr::rom_kernel"]), flush=True)
    except Exception as e:  # noqa: BLE001
        print(name, "failed:", e, r.stderr[-500:], flush=True)
