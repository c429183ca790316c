"""examples/sweep_example.cc is a C++ caller of the C ABI (no Python, no ifopt): the test compiles it against
include/towr_amd.h, runs a 64-candidate Stairs sweep (BASELINE C4) and checks its decision against the same sweep
driven through the ctypes mirror."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import towr_amd as ta
from towr_amd import sweep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build():
    out = os.path.join(ROOT, "examples", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "sweep_example")
    cmd = [HIPCC, "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "sweep_example.cc"), "-L", os.path.join(ROOT, "towr_amd"), "-ltowr_amd",
           "-Wl,-rpath," + os.path.join(ROOT, "towr_amd"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return exe


def test_cpp_example_compiles():
    """(CPU tier: hipcc compiles and links the example against the C ABI without a GPU)"""
    ta.lib()
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_example_sweep_agrees_with_ctypes_path():
    import torch

    exe = _build()
    r = subprocess.run([exe, "64"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"best candidate (\d+) score ([0-9.eE+-]+)", r.stdout)
    assert m, r.stdout
    best_cpp, score_cpp = int(m.group(1)), float(m.group(2))
    states = re.findall(r"^\s+t [0-9.]+ for [0-9.]+ s  contact/plane:((?: -?\d/-?\d+)+)$", r.stdout, re.M)
    assert len(states) >= 2

    model = ta.model_preset("anymal", "stairs")
    cands = sweep.enumerate_candidates(64)
    structs = sweep.candidate_structures(model, cands)
    batch = ta.Batch(structs, list(range(64)), device=0)
    ee0 = [[0.34, 0.19, 0], [0.34, -0.19, 0], [-0.34, 0.19, 0], [-0.34, -0.19, 0]]
    x = np.concatenate([s.initial_guess([0, 0, 0.5], [0, 0, 0], [2.0, 0, 0.5], [0, 0, 0], ee0) for s in structs])
    xd = torch.from_numpy(x).cuda()
    g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
    sc = torch.empty((64, 16), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    batch.eval_device(xd.data_ptr(), g.data_ptr(), 0, ta.EVAL_VALUES, st)
    batch.score_device(g.data_ptr(), sc.data_ptr(), st)
    torch.cuda.synchronize()
    s = sc.cpu().numpy()
    total = s[:, 0] + s[:, 2] + s[:, 6] + s[:, 8]
    best = int(np.argmin(total))
    assert best == best_cpp and abs(total[best] - score_cpp) <= 1e-9 * max(1.0, abs(score_cpp))
    # every foot in contact in the printed plan sits on one of the three regions (index 0..2), feet in the air are -1
    for line in states:
        for pair in line.split():
            c, p = pair.split("/")
            assert (int(c) == 1 and int(p) in (0, 1, 2)) or (int(c) == 0 and int(p) == -1), line
