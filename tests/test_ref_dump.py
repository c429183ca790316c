"""the CMake-gated recipe that compiles the REAL reference sources (oracle/ref_dump/) and
dumps g / Jacobian triplets.  Needs Eigen3 + ifopt, which this image lacks: the recipe must then say "unavailable"
(and the oracle stays pinned by the symbolic, mpmath and known-answer tests).  Where it does build, the dump is
compared with the oracle on the hopper and on the ANYmal trot."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def test_reference_build_recipe_reports_its_availability(tmp_path):
    if not os.path.isdir("/root/reference/towr/src"):
        pytest.skip("the reference sources are not on this box")
    r = subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref_dump", "build.sh")], capture_output=True, text=True, timeout=1200)
    status = open(os.path.join(REF, "STATUS")).read().strip()
    assert status == "available" or status.startswith("unavailable:"), r.stdout + r.stderr
    if status != "available":
        assert "Eigen3" in status or "ifopt" in status or "Could" in status, status
        pytest.skip("reference-native check " + status)
    from oracle import binding as ob

    for robot, terrain, combo, T, mask, n_ee in ((0, 0, 2, 2.0, 63, 1), (3, 3, 1, 2.0, 27, 4)):
        prefix = str(tmp_path / ("r%d" % robot))
        subprocess.check_call([os.path.join(REF, "ref_dump"), str(robot), str(terrain), str(combo), str(T), str(mask), "guess", "1.0", prefix])
        x = np.loadtxt(prefix + "_x.txt")
        g = np.loadtxt(prefix + "_g.txt")
        trip = np.loadtxt(prefix + "_jac.txt")
        P = ob.OracleProblem(robot, terrain, *ob.gait(n_ee, combo, T), constraint_sets=mask)
        og, rp, ci, ov = P.eval(x)
        assert g.shape == og.shape and np.abs(g - og).max() <= 1e-9 * max(1.0, np.abs(og).max())
        rows = np.repeat(np.arange(P.m), np.diff(rp))
        assert np.array_equal(trip[:, 0].astype(int), rows) and np.array_equal(trip[:, 1].astype(int), ci)
        assert np.abs(trip[:, 2] - ov).max() <= 1e-9 * max(1.0, np.abs(ov).max())
