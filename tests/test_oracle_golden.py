"""The CPU oracle (oracle/ca_oracle.cpp) replayed against vectors recorded from the UNMODIFIED reference
(tests/golden/, made by oracle/gen_golden.py).  Two modes per episode:
  * free-running: reset once from the fixture case, step to the end, compare every step;
  * re-injected : load the reference's state at step t, take ONE step, compare with step t+1.
Bars: masks (flags / done / game_over / num_other_agents) bit-exact; float64 quantities within 1e-9 absolute
(north-star bar; a last-bit libm-vs-numpy difference can flip the float32 rounding of an action, a 6e-8 relative jump)."""
import numpy as np
import pytest

from oracle import ca_oracle as orc
from tests import golden_util as gu

TOL = 1e-5
SORT = {"closest_first": orc.SORT_CLOSEST_FIRST, "closest_last": orc.SORT_CLOSEST_LAST,
        "time_to_impact": orc.SORT_TIME_TO_IMPACT}
MASK = 0x3F  # the six agent-state flags
# mixed5 holds a UnicycleDynamicsMaxTurnRate agent: under numpy>=2 (NEP 50) the reference evaluates
# `action[1]/dt` (UnicycleDynamicsMaxTurnRate.py:31) in float32 because action[1] is an np.float32 scalar and dt
# a Python float; under the numpy 1.x rules the code was written for it is float64.  The oracle (and the
# product) follow the float64 reading, so that scenario is held to 1e-7 instead of 1e-12.
REINJECT_TOL = {"mixed5": 1e-7}


def make_oracle(meta, ep):
    over = orc.OVER_ALL_DONE if meta["evaluate"] else orc.OVER_LEARNING_DONE
    p = orc.default_params(1, ep.N, max_obs=meta["K"], dt=meta["dt"], max_time_ratio=meta["max_time_ratio"],
                           sort_mode=SORT[meta["sort"]], game_over_mode=over, rvo_max_neighbors=meta["n_max"])
    o = orc.Oracle(p)
    o.s["policy"][:] = ep.policy
    o.s["dynamics"][:] = ep.dynamics
    learn = ep.policy == orc.POL_LEARNING
    o.s["flags"][:] = np.where(learn, orc.IS_LEARNING | orc.STILL_LEARNING, 0)
    return o


def inject(o, ep, t):
    for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "radius", "pref_speed",
              "time_remaining", "t", "slt"):
        o.s[n][:] = ep.col(t, n)
    o.s["last_action"][:, 0] = ep.col(t, "act0")
    o.s["last_action"][:, 1] = ep.col(t, "act1")
    o.s["flags"][:] = ep.flags[t]
    o.s["step_num"][:] = ep.col(t, "step_num").astype(np.int32)


def check_step(o, ep, t, tol):
    """oracle state/outputs after its step == reference record t+1"""
    for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "time_remaining", "t"):
        np.testing.assert_allclose(o.s[n], ep.col(t + 1, n), rtol=0, atol=tol, err_msg="%s @%d" % (n, t))
    assert np.array_equal(o.s["flags"] & MASK, ep.flags[t + 1] & MASK), "flags @%d" % t
    assert np.array_equal(o.done[0], ep.done[t]), "done @%d" % t
    assert bool(o.game_over[0]) == bool(ep.game_over[t]), "game_over @%d" % t
    assert np.array_equal(o.obs[0][:, 1], ep.obs[t + 1][:, 1]), "num_other_agents @%d" % t
    np.testing.assert_allclose(o.obs[0], ep.obs[t + 1], rtol=0, atol=tol, err_msg="obs @%d" % t)
    np.testing.assert_allclose(o.rewards[0], ep.rewards[t], rtol=0, atol=tol, err_msg="reward @%d" % t)
    moved = (ep.flags[t] & (orc.AT_GOAL | orc.OUT_OF_TIME | orc.IN_COLLISION)) == 0
    # the float32 action pair: 1 float32 ulp at |a| <= 2 is 2.4e-7 (an atan2 that differs in the last float64
    # bit can move the float32 rounding of the action)
    np.testing.assert_allclose(o.s["last_action"][moved, 0], ep.col(t + 1, "act0")[moved], rtol=0, atol=2.5e-7)
    np.testing.assert_allclose(o.s["last_action"][moved, 1], ep.col(t + 1, "act1")[moved], rtol=0, atol=2.5e-7)


@pytest.mark.parametrize("name", gu.SCENARIOS)
def test_free_running_episode(name):
    meta, eps = gu.load(name)
    for c, ep in eps.items():
        o = make_oracle(meta, ep)
        cases, head = ep.case()
        o.reset(cases[None], headings=head[None])
        np.testing.assert_allclose(o.obs[0], ep.obs[0], rtol=0, atol=1e-12, err_msg="reset obs")
        np.testing.assert_allclose(o.s["time_remaining"], ep.col(0, "time_remaining"), rtol=0, atol=1e-12)
        for t in range(ep.T):
            o.step(ep.ext[t][None])
            check_step(o, ep, t, TOL)


@pytest.mark.parametrize("name", gu.SCENARIOS)
def test_reinjected_single_steps(name):
    meta, eps = gu.load(name)
    for c, ep in eps.items():
        o = make_oracle(meta, ep)
        for t in range(ep.T):
            inject(o, ep, t)
            o.step(ep.ext[t][None])
            check_step(o, ep, t, REINJECT_TOL.get(name, 1e-12))


def test_round2_matches_numpy_scalar_round():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-3, 30, 20000), np.arange(-200, 3000) / 100.0 + 0.005,
                         np.array([0.285, 0.575, 1.005, 2.675, 0.125, -0.001, 1.115, 0.0, -0.0])])
    for x in xs:
        assert orc.round2(x) == float(round(np.float64(x), 2))


def test_laserscan_map_and_wall_collisions():
    """Map rasterisation + LaserScanSensor ray-march + static-obstacle collisions against the reference
    (tests/golden/laser4.npz: 4 agents, 3 obstacles, 90 steps, 'laserscan' in the observation)."""
    meta, eps = gu.load("laser4")
    ep = eps[0]
    assert ep.laser.shape == (ep.T + 1, 4, 3, 512) and ep.static_map.any()
    o = make_oracle(meta, ep)
    o.set_map(ep.static_map)
    cases, head = ep.case()
    o.reset(cases[None], headings=head[None])
    o.laserscan()
    mism = 0
    def cmp(t):
        got = np.rint(o.scan[0] / 0.1).astype(np.uint8)
        return int((got != ep.laser[t]).sum())
    mism += cmp(0)
    for t in range(ep.T):
        o.step(ep.ext[t][None])
        check_step(o, ep, t, TOL)
        o.laserscan()
        mism += cmp(t + 1)
    # a beam sample that lands within 1 ulp of a cell edge may floor differently (np.cos vs libm cos)
    assert mism <= 3, mism
    assert (ep.flags[-1] & orc.IN_COLLISION).any(), "the scenario is meant to contain a wall collision"
    assert ep.laser.min() < 60 and (ep.laser == 60).any()
