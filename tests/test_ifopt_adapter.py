"""BASELINE config 1 (towr/test/hopper_example.cc: monoped hop, flat terrain, CPU-Ipopt plumbing) through the ifopt
adapter.  ifopt and Eigen are absent from this image, so towr_amd/csrc/ifopt_adapter.h is compiled against
tests/ifopt_stub/ -- stand-ins for exactly the surface SURVEY.md App. C lists (test infrastructure) -- and driven the
way ifopt::Problem drives constraint sets for Ipopt (tests/ifopt_stub/hopper_adapter_test.cc)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "ifopt_stub")
EXE = os.path.join(ROOT, "tests", "ifopt_stub", "_build", "hopper_adapter_test")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(STUB, "hopper_adapter_test.cc")
    deps = [src, os.path.join(ROOT, "towr_amd", "csrc", "ifopt_adapter.h"), os.path.join(ROOT, "include", "towr_amd.h"),
            os.path.join(ROOT, "towr_amd", "libtowr_amd.so")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + STUB,
                           "-I" + os.path.join(ROOT, "include"), "-o", EXE, src, "-L" + os.path.join(ROOT, "towr_amd"),
                           "-ltowr_amd", "-Wl,-rpath," + os.path.join(ROOT, "towr_amd")])


def test_adapter_compiles_warning_free_and_fails_loudly_without_a_gpu():
    import torch

    _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    r = subprocess.run([EXE, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "no HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("sets,terrain,mode",
                         [(63, "flat", "poll"), (127, "flat", "poll"), (27, "flat", "poll"), (63, "gridmap", "poll"), (127, "gridmap", "poll"),
                          (63, "flat", "push"), (127, "gridmap", "push"), (63, "flat", "strict")],
                         ids=["towr_default", "optimised_timings", "hot_path", "grid_map_terrain", "grid_map_optimised_timings",
                              "towr_default_pushed", "grid_map_optimised_timings_pushed", "towr_default_polled_on_every_request"])
def test_hopper_through_the_ifopt_surface(sets, terrain, mode):
    """MakeDeviceConstraints -> GetValues / GetBounds / GetJacobian of every set == twr_batch_eval_host, stacked; on flat
    ground (hopper_example.cc) and on the `Grid` terrain fpowr hands the solver (footstep_plan_server.cc:155); an unknown
    variable-set name must throw, a missing one too.  The variable sets are towr-shaped stand-ins that count their GetValues
    calls: x is read once per new x -- pushed by observers like the reference's (nodes_variables.cc:64-79), or polled once
    per sweep -- never once per GetValues / FillJacobianBlock request."""
    _build()
    args = [EXE, "--gpu", str(sets), terrain, mode]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max|dg|=0 max|dJ|=0" in r.stdout and " ok" in r.stdout, r.stdout
    assert ("grid_map terrain" in r.stdout) == (terrain == "gridmap")
    assert "x-change detection: " + {"poll": "poll per sweep", "strict": "poll on every request", "push": "push"}[mode] in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["push", "poll", "strict"])
def test_quadruped_default_list_host_cost_per_iteration(mode):
    """ANYmal, towr's default list: 19 constraint sets x 10 variable sets = 209 requests per Ipopt iteration.
    The adapter reads x at most once per variable set and SetVariables (push) / sweep (polled), evaluates values once and
    the Jacobian once per iteration; the binary checks the read count, this test the shape of its report."""
    import json

    _build()
    r = subprocess.run([EXE, "--quadruped", mode, "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["n_con_sets"] == 19 and rep["n_var_sets"] == 10 and rep["requests_per_iteration"] == 209
    assert rep["value_evals"] == 40 and rep["jacobian_evals"] == 40
    if mode != "strict":
        assert rep["variable_set_reads_per_iteration"] <= (20.0 if mode == "push" else 30.0)
        # the host side of the boundary stays well below what re-reading x on every request cost (round 4)
        assert rep["host_change_detection_us_per_iteration"] < 0.25 * rep["round4_rule_us_per_iteration"]
