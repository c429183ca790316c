"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on identical inputs."""
import numpy as np
import pytest

import towr_amd as ta
from tests.common import Case, assert_parity, baseline_cases, hopper_schedule, k_params

pytestmark = pytest.mark.gpu


def _eval_case(case, xs):
    batch = ta.Batch([case.S], [0] * len(xs), device=0)
    g, j = batch.eval_host(np.concatenate(xs))
    return batch, g, j


@pytest.mark.parametrize("name", list(baseline_cases().keys()))
def test_baseline_configs_match_oracle(name):
    case = baseline_cases()[name]()
    goal = 2.0 if case.terrain != "flat" else 1.0
    xs = [case.x_guess(goal), case.x_perturbed(0, goal), case.x_perturbed(1, goal), case.x_wild(0)]
    batch, g, j = _eval_case(case, xs)
    for p, x in enumerate(xs):
        rg, _, _, rj = case.P.eval(x)
        assert_parity(case.S, g[batch.g_off[p]:batch.g_off[p + 1]], j[batch.jac_off[p]:batch.jac_off[p + 1]], rg, rj,
                      "%s x[%d]" % (name, p))
