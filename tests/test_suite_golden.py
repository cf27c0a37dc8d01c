"""The reference's own benchmark -- its full test suite (experiments/src/run_full_test_suite.py:54-130: run_episode over
the 500 fixture cases, all agents RVO) -- recorded from the unmodified reference (oracle/gen_suite_golden.py ->
tests/golden/suite_*.npz) against the CPU oracle: per-case outcome, step count, time to goal, total reward, final flags
and positions.  The oracle runs on the same libm (glibc) as the reference's numpy, so EVERY case must agree, including
the symmetric head-on cases whose paths hinge on the last bit of atan2.  `ragged4` holds 2-, 3- and 4-agent episodes in
4-slot envs (the reference's per-episode agent count, test_cases.py:224-227)."""
import numpy as np
import pytest

from tests import golden_util as gu


@pytest.mark.parametrize("name", ["n10", "n4", "ragged4"])
def test_oracle_reproduces_the_reference_suite(name):
    from oracle import ca_oracle as orc
    ref = gu.load_suite(name)
    cases = gu.suite_cases(name)
    E, N = cases.shape[:2]
    o = orc.Oracle(orc.default_params(E, N, ragged=int(name == "ragged4")))
    o.s["policy"][:] = orc.POL_RVO
    o.reset(cases)
    if name == "ragged4":
        assert np.array_equal(((o.view("flags") >> 16) & 1).sum(1), 4 - ref["num_agents"])

    def step():
        o.step()
        return o.game_over, {k: o.view(k) for k in ("t", "slt", "ep_reward", "pos_x", "pos_y")}

    got = gu.run_suite(o, cases, lambda: o.view("flags"), step)
    assert np.array_equal(got["outcome"], ref["outcome"])
    assert np.array_equal(got["steps"], ref["steps"])
    assert np.array_equal(got["flags"], ref["flags"])
    np.testing.assert_allclose(got["time_to_goal"], ref["time_to_goal"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got["extra_time_to_goal"], ref["extra_time_to_goal"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got["total_reward"], ref["total_reward"], rtol=0, atol=1e-4)   # (free-running sums over up to 2000 steps)
    np.testing.assert_allclose(got["pos"], ref["pos"], rtol=0, atol=1e-4)
