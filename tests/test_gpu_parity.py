"""GPU parity: the HIP path (through the C ABI, libcagpu.so) against the CPU oracle and the golden vectors
recorded from the unmodified reference.  Bars (BASELINE.json north_star): collision / done masks bit-exact,
positions / observations within 1e-5."""
import ctypes
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests import golden_util as gu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

TOL = 1e-5
MASK = 0x3F
SORT = {"closest_first": 0, "closest_last": 1, "time_to_impact": 2}
F64 = ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "radius", "pref_speed",
       "time_remaining", "t", "slt", "ep_reward", "turning_dir")


def _mods():
    from gym_collision_avoidance_amd import _native as nat
    from gym_collision_avoidance_amd import core
    from oracle import ca_oracle as orc
    return nat, core, orc


def _pair(E, N, K=None, **kw):
    """An oracle and a GPU sim with identical parameters."""
    nat, core, orc = _mods()
    po = orc.default_params(E, N, max_obs=K, **{k: v for k, v in kw.items()})
    pg = core.make_params(E, N, max_obs=K, **{("near_goal_threshold" if k == "near_goal" else
                                               "getting_close_range" if k == "getting_close" else k): v
                                              for k, v in kw.items()})
    return orc.Oracle(po), core.BatchedSim(pg, record_actions=True)


def _upload(o, g):
    """GPU state := oracle state (re-injection)."""
    nat, core, orc = _mods()
    for n in F64:
        g.state[n].copy_(torch.from_numpy(o.s[n].reshape(o.E, o.N)))
    g.state["last_action"].copy_(torch.from_numpy(o.s["last_action"].reshape(o.E, o.N, 2)))
    fl = (o.s["flags"].astype(np.int64) & (0xFF | orc.ABSENT)) | (o.s["policy"].astype(np.int64) << nat.POLICY_SHIFT) | \
         (o.s["dynamics"].astype(np.int64) << nat.DYNAMICS_SHIFT)      # (PLAN_VALID clear: the plan is forgotten)
    g.state["flags"].copy_(torch.from_numpy(fl.astype(np.int32).reshape(o.E, o.N)))
    g.state["step_num"].copy_(torch.from_numpy(o.s["step_num"].reshape(o.E, o.N)))
    g.state["episode_step"].copy_(torch.from_numpy(o.s["episode_step"]))
    g.state["reset_count"].copy_(torch.from_numpy(o.s["reset_count"]))
    g.state["env_stats"].copy_(torch.from_numpy(o.s["env_stats"]))


def _download(g, o):
    """oracle state := GPU state (the reverse re-injection: both then continue from IDENTICAL bits)."""
    nat, core, orc = _mods()
    for n in F64:
        o.s[n][:] = g.state[n].cpu().numpy().reshape(-1)
    o.s["last_action"][:] = g.state["last_action"].cpu().numpy().reshape(o.s["last_action"].shape)
    fl = g.state["flags"].cpu().numpy().reshape(-1).astype(np.uint32)
    o.s["flags"][:] = (fl & 0xFF) | (fl & orc.ABSENT)
    o.s["step_num"][:] = g.state["step_num"].cpu().numpy().reshape(-1)
    o.s["episode_step"][:] = g.state["episode_step"].cpu().numpy()
    o.s["reset_count"][:] = g.state["reset_count"].cpu().numpy()
    o.s["env_stats"][:] = g.state["env_stats"].cpu().numpy()


def _compare(o, g, tol=TOL, what=""):
    """masks exact, floats within tol"""
    gs = {n: g.state[n].cpu().numpy().reshape(-1) for n in F64}
    gf = g.state["flags"].cpu().numpy().reshape(-1).astype(np.uint32)
    assert np.array_equal(gf & MASK, o.s["flags"] & MASK), "flags " + what
    assert np.array_equal(g.done.cpu().numpy(), o.done), "done " + what
    assert np.array_equal(g.game_over.cpu().numpy(), o.game_over), "game_over " + what
    for n in F64:
        if n == "heading":  # an angle: at the wrap boundary of [-pi, pi) a last-bit libm difference picks the other
            # representative of the SAME heading (seen with agents driving exactly along -x), so compare modulo 2 pi
            d = np.abs((gs[n] - o.s[n] + np.pi) % (2 * np.pi) - np.pi)
            assert d.max() <= tol, "heading %s: %g" % (what, d.max())
            continue
        if n == "turning_dir":  # branches on the SIGN of the new heading (UnicycleDynamics.py:41-47): an agent whose heading
            # sits at the wrap boundary (+-pi: see above) or at 0 within an ulp may take the other branch -- a handful of agents
            # in the fuzzed configurations, none in the reference-recorded episodes (tests/test_gpu_env_api.py)
            bad = np.abs(gs[n] - o.s[n]) > tol
            h = o.s["heading"]
            on_edge = (np.abs(h) < 1e-7) | (np.abs(np.abs(h) - np.pi) < 1e-7) | \
                      (np.abs(gs["heading"]) < 1e-7) | (np.abs(np.abs(gs["heading"]) - np.pi) < 1e-7)
            assert not (bad & ~on_edge).any(), "turning_dir %s: %d of %d differ away from a sign change of the heading" % (
                what, (bad & ~on_edge).sum(), bad.size)
            assert bad.mean() <= 0.01, "turning_dir %s: %d of %d differ" % (what, bad.sum(), bad.size)
            continue
        np.testing.assert_allclose(gs[n], o.s[n], rtol=0, atol=tol, err_msg=n + " " + what)
    gobs = g.obs.cpu().numpy().astype(np.float64)
    assert np.array_equal(gobs[..., 1], o.obs[..., 1]), "num_other_agents " + what
    np.testing.assert_allclose(gobs, o.obs, rtol=0, atol=tol, err_msg="obs " + what)
    np.testing.assert_allclose(g.rewards.cpu().numpy(), o.rewards, rtol=0, atol=tol, err_msg="rewards " + what)
    assert np.array_equal(g.state["step_num"].cpu().numpy().reshape(-1), o.s["step_num"]), "step_num " + what
    assert np.array_equal(g.state["episode_step"].cpu().numpy(), o.s["episode_step"]), "episode_step " + what
    assert np.array_equal(g.state["reset_count"].cpu().numpy(), o.s["reset_count"]), "reset_count " + what


def test_library_loads_on_gpu():
    nat, core, orc = _mods()
    assert nat.lib().cagpu_version() == nat.ABI_VERSION
    assert torch.cuda.is_available()


# ---------------------------------------------------------------- golden vectors (reference-recorded)
def _golden_sim(meta, ep):
    nat, core, orc = _mods()
    over = nat.OVER_ALL_DONE if meta["evaluate"] else nat.OVER_LEARNING_DONE
    p = core.make_params(1, ep.N, max_obs=meta["K"], dt=meta["dt"], max_time_ratio=meta["max_time_ratio"],
                         sort_mode=SORT[meta["sort"]], game_over_mode=over, rvo_max_neighbors=meta["n_max"])
    gu.apply_constants(meta, p)
    g = core.BatchedSim(p)
    g.set_plugins(ep.policy[None], ep.dynamics[None])
    return g


def _check_golden_step(g, ep, t, tol, tol_heading=None):
    tol_heading = tol if tol_heading is None else tol_heading
    for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "time_remaining", "t"):
        np.testing.assert_allclose(g.state[n].cpu().numpy()[0], ep.col(t + 1, n), rtol=0,
                                   atol=tol_heading if n in ("heading", "vel_x", "vel_y") else tol,
                                   err_msg="%s @%d" % (n, t))
    gf = g.state["flags"].cpu().numpy()[0].astype(np.uint32)
    assert np.array_equal(gf & MASK, ep.flags[t + 1] & MASK), "flags @%d" % t
    assert np.array_equal(g.done.cpu().numpy()[0], ep.done[t]), "done @%d" % t
    assert bool(g.game_over.cpu().numpy()[0]) == bool(ep.game_over[t]), "game_over @%d" % t
    gobs = g.obs.cpu().numpy()[0].astype(np.float64)
    assert np.array_equal(gobs[:, 1], ep.obs[t + 1][:, 1]), "num_other_agents @%d" % t
    np.testing.assert_allclose(gobs, ep.obs[t + 1], rtol=0, atol=tol_heading, err_msg="obs @%d" % t)
    np.testing.assert_allclose(g.rewards.cpu().numpy()[0], ep.rewards[t], rtol=0, atol=tol, err_msg="reward @%d" % t)


@pytest.mark.parametrize("name", gu.SCENARIOS)
def test_golden_free_running(name):
    """reset once from the recorded case, step to the end of the episode, compare every step.  Free-running
    over 50-370 steps is an ill-conditioned comparison (the heading of an agent that ORCA has nearly stopped is
    atan2 of a ~1e-3 displacement, so a last-bit difference grows to ~1e-5 in heading): positions are held to
    1e-4, headings / velocities / ego-frame observations to 1e-3, masks exact.  The strict 1e-5 bar is applied
    per step in test_golden_reinjected."""
    meta, eps = gu.load(name)
    for c, ep in eps.items():
        g = _golden_sim(meta, ep)
        cases, head = ep.case()
        g.reset(cases[None], headings=head[None])
        np.testing.assert_allclose(g.obs.cpu().numpy()[0], ep.obs[0], rtol=0, atol=TOL)
        for t in range(ep.T):
            g.step(ep.ext[t][None])
            _check_golden_step(g, ep, t, 1e-4, 1e-3)


@pytest.mark.parametrize("name", gu.SCENARIOS)
def test_golden_reinjected(name):
    """load the reference's state at step t, take ONE step, compare with its step t+1"""
    nat, core, orc = _mods()
    meta, eps = gu.load(name)
    for c, ep in eps.items():
        g = _golden_sim(meta, ep)
        for t in range(ep.T):
            for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "radius", "pref_speed",
                      "time_remaining", "t", "slt"):
                g.state[n].copy_(torch.from_numpy(ep.col(t, n)[None].copy()))
            la = np.stack([ep.col(t, "act0"), ep.col(t, "act1")], axis=1).astype(np.float32)
            g.state["last_action"].copy_(torch.from_numpy(la[None]))
            g.state["flags"].copy_(torch.from_numpy((ep.flags[t] & 0xFF).astype(np.int32)[None]))
            g.set_plugins(ep.policy[None], ep.dynamics[None])
            g.step(ep.ext[t][None])
            _check_golden_step(g, ep, t, 1e-6)


# ---------------------------------------------------------------- against the oracle at scale
# (10, 5200, 25): more workgroups than fit with the LDS staging area -> the launcher picks the unstaged layout
@pytest.mark.parametrize("N,E,steps", [(10, 600, 160), (4, 333, 120), (2, 100, 60), (3, 64, 60), (10, 5200, 25)])
def test_reinjected_vs_oracle_fixtures(N, E, steps):
    """fixture cases, RVO, auto-reset; every step the GPU restarts from the oracle's state"""
    nat, core, orc = _mods()
    table = gu.fixtures(N)
    o, g = _pair(E, N)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    cases = table[np.arange(E) % 500]
    o.reset(cases)
    g.reset(cases)
    _compare_reset(o, g)
    for t in range(steps):
        _upload(o, g)
        o.rollout(table, 1)
        g.step()
        _compare(o, g, what="N=%d step %d" % (N, t))
        np.testing.assert_allclose(g.state["env_stats"].cpu().numpy(), o.s["env_stats"], rtol=0, atol=1e-6)
    assert o.s["env_stats"][:, 0].sum() > 0 or steps < 100  # some episodes ended -> auto-reset path exercised


@pytest.mark.parametrize("N,E,K", [(6, 200, 4), (10, 300, 9), (20, 40, 19)])
def test_time_to_impact_sorting_vs_oracle(N, E, K):
    """agent_sorting_method = time_to_impact (util.py:23-127), re-injected each step, mixed velocities"""
    nat, core, orc = _mods()
    rng = np.random.default_rng(N)
    o, g = _pair(E, N, K, sort_mode=2)
    pol = np.where(rng.random((E, N)) < 0.7, orc.POL_RVO, orc.POL_NONCOOP).astype(np.int32)
    o.s["policy"][:] = pol.reshape(-1)
    g.set_plugins(pol)
    cases = np.zeros((E, N, 6))
    cases[..., 0:2] = rng.uniform(-6, 6, (E, N, 2))
    cases[..., 2:4] = rng.uniform(-6, 6, (E, N, 2))
    cases[..., 4] = rng.uniform(0.5, 2.0, (E, N))
    cases[..., 5] = rng.uniform(0.2, 0.6, (E, N))
    o.reset(cases)
    g.reset(cases)
    _compare_reset(o, g)
    for t in range(60):
        _upload(o, g)
        o.step()
        g.step()
        _compare(o, g, what="tti N=%d step %d" % (N, t))


def _compare_reset(o, g):
    gobs = g.obs.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(gobs, o.obs, rtol=0, atol=TOL)
    for n in F64:
        np.testing.assert_allclose(g.state[n].cpu().numpy().reshape(-1), o.s[n], rtol=0, atol=1e-9, err_msg=n)


def _swap_cases(table):
    """fixture cases holding a pair of agents that exchange places exactly (start_i == goal_j and start_j == goal_i):
    the two meet head-on on one line, a perfectly symmetric ORCA problem whose resolution is decided by round-off"""
    s, g = table[:, :, 0:2], table[:, :, 2:4]
    a = np.abs(s[:, :, None, :] - g[:, None, :, :]).max(axis=-1) < 1e-9       # [C, i, j]: start_i == goal_j
    both = a & np.swapaxes(a, 1, 2)
    both &= ~np.eye(table.shape[1], dtype=bool)[None]
    return both.any(axis=(1, 2))


def test_free_running_vs_oracle_10_agents():
    """no re-injection: 256 envs x 10 agents x 200 steps with auto-reset.  This comparison is ill-posed for the
    fixture cases in which two agents swap places exactly (71 of the 500 ten-agent cases): they meet head-on in a
    perfectly symmetric configuration, which ORCA resolves by amplifying lateral round-off noise ~12x per step
    (measured: 4e-16 -> 2e-5 in 11 steps, in the CPU oracle as much as on the GPU).  The seed of that noise is the
    last bit of atan2 at +-pi, where ROCm's libm and glibc differ, so those envs may legitimately take different
    (mirror-image) paths.  Bar: every env that disagrees has been through such a swap case; every other env agrees in
    masks exactly and in position within 1e-5; and the two populations' episode statistics stay within a few episodes."""
    nat, core, orc = _mods()
    N, E, steps = 10, 256, 200
    table = gu.fixtures(N)
    swap = _swap_cases(table)
    assert 50 < swap.sum() < 100
    o, g = _pair(E, N)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    cases = table[np.arange(E) % 500]
    o.reset(cases)
    g.reset(cases)
    o.rollout(table, steps)
    for _ in range(steps):
        g.step()
    gf = g.state["flags"].cpu().numpy().reshape(E, N).astype(np.uint32) & MASK
    of = (o.s["flags"] & MASK).reshape(E, N)
    env_ok = (gf == of).all(axis=1)
    pos_err = np.maximum(np.abs(g.state["pos_x"].cpu().numpy() - o.s["pos_x"].reshape(E, N)).max(axis=1),
                         np.abs(g.state["pos_y"].cpu().numpy() - o.s["pos_y"].reshape(E, N)).max(axis=1))
    # cases env e has been through so far (either run): (e + k E) % 500 for k = 0 .. resets taken
    resets = np.maximum(o.s["reset_count"], g.state["reset_count"].cpu().numpy())
    excused = np.array([swap[(e + np.arange(resets[e] + 1) * E) % 500].any() for e in range(E)])
    diverged = ~env_ok | ~(pos_err < TOL)
    assert not (diverged & ~excused).any(), "envs %s diverged without having met a swap case (pos err %s)" % (
        np.nonzero(diverged & ~excused)[0][:10], pos_err[diverged & ~excused][:10])
    assert env_ok.mean() >= 0.9 and (pos_err < TOL).mean() >= 0.9, (env_ok.mean(), (pos_err < TOL).mean())
    gs, os_ = g.episode_stats().cpu().numpy(), o.s["env_stats"].sum(axis=0)
    assert abs(gs[0] - os_[0]) <= 4 and abs(gs[1] - os_[1]) <= 4


@pytest.mark.parametrize("E,T", [(100, 150), (5000, 40)])      # 5000 envs: the crowded launch geometry
def test_rollout_equals_repeated_steps(E, T):
    nat, core, orc = _mods()
    N = 10
    table = gu.fixtures(N)
    sims = []
    for _ in range(2):
        g = core.BatchedSim(core.make_params(E, N))
        g.set_plugins(nat.POL_RVO)
        g.set_fixture_table(table)
        g.reset_from_table()
        sims.append(g)
    for _ in range(T):
        sims[0].step()
    sims[1].rollout(T)
    for n in F64 + ("flags", "step_num", "episode_step", "reset_count", "env_stats", "last_action"):
        assert torch.equal(sims[0].state[n], sims[1].state[n]), n
    assert torch.equal(sims[0].obs, sims[1].obs)
    assert torch.equal(sims[0].rewards, sims[1].rewards)
    assert torch.equal(sims[0].done, sims[1].done)


# ---------------------------------------------------------------- the metric geometry itself (4096 x 10 and its neighbours)
def _pipe_tile(N, E):
    """envs per tile of the pipelined kernel (launch_pipe, csrc/cagpu.hip): 4 for N = 10 at every batch size (1- / 2- /
    3-env tiles for small batches measured slower in round 4), floor(64 / N) for the other agent counts"""
    return 4 if N == 10 else 64 // N


def _expected_kernel(E, multi, pipeline=True):
    """the instantiation the launcher must pick on a 256-CU MI355X for N = 10 with precomputed reset observations: the
    software-pipelined kernel (CaState.next_action given) while every workgroup is resident at once; otherwise
    4-env tiles while ceil(E / 4) <= 4 x CUs; the staged observation block only while <= 3 workgroups per CU"""
    wgs4 = (E + 3) // 4
    if pipeline and (wgs4 <= 4 * 256 or wgs4 >= 8 * 256):
        return "ca_pipe_kernel<10, %d, %s>" % (_pipe_tile(10, E), "true" if multi else "false")
    te = 4 if wgs4 <= 4 * 256 else 0
    wgs = wgs4 if te == 4 else (E + 5) // 6
    stage = wgs <= 3 * 256
    return "ca_kernel<256, %s, 10, %s, true, %d>" % ("true" if stage else "false", "true" if multi else "false", te)


@pytest.mark.parametrize("E", [3073, 4095, 4096])
def test_metric_geometry_step_vs_oracle(E):
    """The exact kernel instantiation bench.py times (4-env tiles, unstaged, N compiled in, precomputed reset
    observations), re-injected from the oracle for 30 steps with auto-reset: masks exact, floats within 1e-5."""
    nat, core, orc = _mods()
    N, steps = 10, 30
    table = gu.fixtures(N)
    o, g = _pair(E, N)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    cases = table[np.arange(E) % 500]
    o.reset(cases)
    g.reset(cases)
    _compare_reset(o, g)
    # start mid-episode so that time-outs / goals / collisions / auto-resets all occur within the compared window
    o.rollout(table, 140)
    for t in range(steps):
        _upload(o, g)
        o.rollout(table, 1)
        g.step()
        assert nat.lib().cagpu_last_kernel().decode().startswith(_expected_kernel(E, False)), nat.lib().cagpu_last_kernel()
        _compare(o, g, what="E=%d step %d" % (E, t))
        np.testing.assert_allclose(g.state["env_stats"].cpu().numpy(), o.s["env_stats"], rtol=0, atol=1e-6)
    assert o.s["env_stats"][:, 0].sum() > E // 8       # plenty of episodes ended inside the window


@pytest.mark.parametrize("E", [3073, 4096])
def test_metric_geometry_rollout_vs_oracle(E):
    """cagpu_rollout (the fused n-step instantiation at the metric geometry) against the ORACLE's rollout: both start
    from the same state, run 3 steps on their own, are compared, and the oracle's state is re-injected (a free run of 3
    steps cannot amplify the libm round-off of a symmetric encounter beyond the 1e-5 bar)."""
    nat, core, orc = _mods()
    N = 10
    table = gu.fixtures(N)
    o, g = _pair(E, N)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    cases = table[np.arange(E) % 500]
    o.reset(cases)
    g.reset(cases)
    o.rollout(table, 140)
    for r in range(10):
        _upload(o, g)
        o.rollout(table, 3)
        g.rollout(3)
        assert nat.lib().cagpu_last_kernel().decode().startswith(_expected_kernel(E, True)), nat.lib().cagpu_last_kernel()
        _compare(o, g, what="E=%d rollout round %d" % (E, r))
        np.testing.assert_allclose(g.state["env_stats"].cpu().numpy(), o.s["env_stats"], rtol=0, atol=1e-6)
    assert o.s["env_stats"][:, 0].sum() > E // 8


def test_bench_kernel_is_the_tested_kernel():
    """bench.py's workload (4096 x 10, fixture table, reset observations precomputed) selects the instantiation the two
    tests above hold against the oracle"""
    nat, core, orc = _mods()
    g = core.BatchedSim(core.make_params(4096, 10))
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(gu.fixtures(10))
    g.reset_from_table()
    g.step()
    assert nat.lib().cagpu_last_kernel().decode().startswith("ca_pipe_kernel<10, 4, false> grid=1024")
    g.rollout(5)
    assert nat.lib().cagpu_last_kernel().decode().startswith("ca_pipe_kernel<10, 4, true> grid=1024")
    h = core.BatchedSim(core.make_params(4096, 10), pipeline=False)     # without next_action: the round-2 kernel
    h.set_plugins(nat.POL_RVO)
    h.set_fixture_table(gu.fixtures(10))
    h.reset_from_table()
    h.step()
    assert nat.lib().cagpu_last_kernel().decode().startswith("ca_kernel<256, false, 10, false, true, 4> grid=1024")


# ---------------------------------------------------------------- ORCA stage alone (rvo2 replacement)
@pytest.mark.parametrize("N,E,spread", [(10, 2000, 6.0), (10, 2000, 1.5), (5, 1000, 0.8), (2, 500, 1.0),
                                        (20, 300, 3.0), (50, 64, 4.0), (64, 16, 6.0)])
def test_orca_bit_exact(N, E, spread):
    """random (also overlapping / infeasible -> linearProgram3) configurations; float results must be bit-identical"""
    nat, core, orc = _mods()
    rng = np.random.default_rng(N * 1000 + E)
    pos = rng.uniform(-spread, spread, (E, N, 2)).astype(np.float32)
    vel = rng.uniform(-1.2, 1.2, (E, N, 2)).astype(np.float32)
    pref = rng.uniform(-1.5, 1.5, (E, N, 2)).astype(np.float32)
    radius = rng.uniform(0.2, 0.8, (E, N)).astype(np.float32)
    ms = rng.uniform(0.5, 1.5, (E, N)).astype(np.float32)
    want = orc.orca(pos, vel, pref, radius, ms)
    dev = "cuda:0"
    got = core.orca(*(torch.from_numpy(x).to(dev) for x in (pos, vel, pref, radius, ms))).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        "max abs diff %g on %d values" % (np.abs(got - want).max(), (got != want).sum())


def test_orca_limited_neighbours():
    nat, core, orc = _mods()
    rng = np.random.default_rng(5)
    E, N = 500, 12
    pos = rng.uniform(-4, 4, (E, N, 2)).astype(np.float32)
    vel = rng.uniform(-1, 1, (E, N, 2)).astype(np.float32)
    pref = rng.uniform(-1, 1, (E, N, 2)).astype(np.float32)
    radius = rng.uniform(0.2, 0.5, (E, N)).astype(np.float32)
    ms = np.full((E, N), 1.0, np.float32)
    want = orc.orca(pos, vel, pref, radius, ms, max_neighbors=4, neighbor_dist=3.0)
    got = core.orca(*(torch.from_numpy(x).cuda() for x in (pos, vel, pref, radius, ms)), max_neighbors=4,
                    neighbor_dist=3.0).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_too_many_agents_is_a_loud_error():
    nat, core, orc = _mods()
    with pytest.raises(nat.CagpuError):
        core.orca(*(torch.zeros(s, device="cuda") for s in ((2, 70, 2), (2, 70, 2), (2, 70, 2), (2, 70), (2, 70))))
    core.BatchedSim(core.make_params(2, 65)).observe()       # (65 .. 1024 agents: the large-env kernel, csrc/cagpu_big.inc)
    assert nat.lib().cagpu_last_kernel().decode().startswith("ca_big_kernel")
    core.BatchedSim(core.make_params(2, 257)).observe()
    assert "threads=512" in nat.lib().cagpu_last_kernel().decode()
    with pytest.raises(nat.CagpuError):
        core.BatchedSim(core.make_params(2, 1025)).observe()  # one thread per agent ends at the largest workgroup


# ---------------------------------------------------------------- static map + LaserScanSensor (config 5 row)
def _laser_idx(scan):
    return np.rint(np.asarray(scan, dtype=np.float64) / 0.1).astype(np.uint8)


def test_laserscan_golden_episode():
    """Map rasterisation, ray-march, history roll and wall collisions against the reference-recorded episode"""
    nat, core, orc = _mods()
    meta, eps = gu.load("laser4")
    ep = eps[0]
    g = _golden_sim(meta, ep)
    g.set_map(ep.static_map)
    cases, head = ep.case()
    g.reset(cases[None], headings=head[None])
    mism = int((_laser_idx(g.laserscan().cpu().numpy()[0]) != ep.laser[0]).sum())
    for t in range(ep.T):
        g.step(ep.ext[t][None])
        _check_golden_step(g, ep, t, 1e-4, 1e-3)
        mism += int((_laser_idx(g.laserscan().cpu().numpy()[0]) != ep.laser[t + 1]).sum())
    assert mism <= 3, mism       # a sample within 1 ulp of a cell edge may floor differently (libm cos / sin)
    assert (g.state["flags"].cpu().numpy() & nat.IN_COLLISION).any()   # the wall collision happened here too


@pytest.mark.parametrize("N,E,with_map", [(4, 50, True), (10, 40, True), (10, 40, False), (50, 6, True)])
def test_laserscan_and_walls_vs_oracle(N, E, with_map):
    """random scenes with random static obstacles, re-injected every step; scans must agree on (practically) every beam"""
    nat, core, orc = _mods()
    rng = np.random.default_rng(N + E)
    o, g = _pair(E, N, min(N - 1, 9))
    pol = np.where(rng.random((E, N)) < 0.5, orc.POL_RVO, orc.POL_NONCOOP).astype(np.int32)
    o.s["policy"][:] = pol.reshape(-1)
    g.set_plugins(pol)
    static = None
    if with_map:
        static = rng.random((160, 160)) < 0.004
        static[70:74, 20:140] = True
    o.set_map(static)
    g.set_map(static)
    cases = np.zeros((E, N, 6))
    cases[..., 0:2] = rng.uniform(-7, 7, (E, N, 2))
    cases[..., 2:4] = rng.uniform(-7, 7, (E, N, 2))
    cases[..., 4] = rng.uniform(0.5, 2.0, (E, N))
    cases[..., 5] = rng.uniform(0.2, 0.8, (E, N))
    cases[0, 0, 0:2] = [9.5, 9.5]      # an agent outside the 16 m x 16 m map: no disc, no ego mask, beams mostly off-map
    o.reset(cases)
    g.reset(cases)
    total = bad = 0
    for t in range(12):
        if t:
            _upload(o, g)
            o.step()
            g.step()
            _compare(o, g, what="laser N=%d step %d" % (N, t))
        want = _laser_idx(o.laserscan())
        got = _laser_idx(g.laserscan().cpu().numpy())
        assert np.array_equal(g.scan_hist.cpu().numpy() == 255, want == 60)
        bad += int((got != want).sum())
        total += want.size
        o.scan_hist[:] = g.scan_hist.cpu().numpy()   # keep the histories in step despite a rare 1-ulp beam
    assert bad <= max(3, total // 200000), "%d of %d beams differ" % (bad, total)
    if with_map:
        assert (o.s["flags"] & orc.IN_COLLISION).any()


# ---------------------------------------------------------------- edge cases
@pytest.mark.parametrize("N,E,K", [(1, 7, 3), (2, 1, 1), (7, 130, 3), (10, 1, 9), (13, 41, 20), (33, 9, 8),
                                   (64, 3, 10), (50, 11, 49)])
def test_shapes_mixed_plugins_vs_oracle(N, E, K):
    """ragged tiles (E not a multiple of the tile), K < N-1 and K > N-1, every built-in policy / dynamics,
    closest_last ordering; re-injected each step"""
    nat, core, orc = _mods()
    rng = np.random.default_rng(N * 100 + E)
    o, g = _pair(E, N, K, sort_mode=1, max_time_ratio=2.0, game_over_mode=2)
    pol = rng.integers(0, 6, (E, N)).astype(np.int32)
    dyn = rng.integers(0, 3, (E, N)).astype(np.int32)
    o.s["policy"][:] = pol.reshape(-1)
    o.s["dynamics"][:] = dyn.reshape(-1)
    learn = (pol == orc.POL_LEARNING) | (pol == orc.POL_LEARNING_GA3C)
    o.s["flags"][:] = np.where(learn, orc.IS_LEARNING | orc.STILL_LEARNING, 0).reshape(-1)
    g.set_plugins(pol, dyn)
    cases = np.zeros((E, N, 6))
    cases[..., 0:2] = rng.uniform(-5, 5, (E, N, 2))
    cases[..., 2:4] = rng.uniform(-5, 5, (E, N, 2))
    cases[..., 4] = rng.uniform(0.5, 2.0, (E, N))
    cases[..., 5] = rng.uniform(0.2, 0.8, (E, N))
    o.reset(cases)
    g.reset(cases)
    _compare_reset(o, g)
    for t in range(40):
        ext = rng.uniform(0, 1, (E, N, 2))
        ext[pol == orc.POL_LEARNING_GA3C, 0] = rng.integers(0, 11, (pol == orc.POL_LEARNING_GA3C).sum())
        _upload(o, g)
        o.step(ext)
        g.step(ext)
        _compare(o, g, what="N=%d step %d" % (N, t))
        np.testing.assert_allclose(g.actions.cpu().numpy(), o.actions, rtol=0, atol=2.5e-7)


def test_masked_reset_and_observe():
    nat, core, orc = _mods()
    N, E = 10, 77
    table = gu.fixtures(N)
    o, g = _pair(E, N)
    g.set_plugins(nat.POL_RVO)
    cases = table[np.arange(E) % 500]
    o.reset(cases)
    g.reset(cases)
    for _ in range(20):
        o.step()
        g.step()
    _upload(o, g)        # re-synchronise (free-running may have drifted on symmetric cases, see above)
    g.observe()          # cagpu_observe: obs of the CURRENT state, no stepping
    np.testing.assert_allclose(g.obs.cpu().numpy(), o.obs, rtol=0, atol=TOL)
    mask = (np.arange(E) % 3 == 0).astype(np.uint8)
    cases2 = table[(np.arange(E) + 100) % 500]
    o.reset(cases2, mask=mask)
    g.reset(cases2, mask=mask)
    _compare_reset(o, g)
    assert np.array_equal(g.state["t"].cpu().numpy()[mask == 1], np.zeros((int(mask.sum()), N)))
    assert (g.state["t"].cpu().numpy()[mask == 0] > 0).all()
    before = g.obs.clone()
    g.obs.zero_()
    g.observe()
    assert torch.equal(before, g.obs)


# ---------------------------------------------------------------- full benchmark size: size-independent properties
def test_full_size_properties():
    """4096 envs x 10 agents (the metric config): invariants that hold for any correct step"""
    nat, core, orc = _mods()
    N, E = 10, 4096
    table = gu.fixtures(N)
    g = core.BatchedSim(core.make_params(E, N))
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    g.reset_from_table()
    prev_t = g.state["t"].clone()
    for it in range(300):
        g.step()
    f = g.state["flags"].cpu().numpy().astype(np.uint32)
    done = (f & (nat.AT_GOAL | nat.OUT_OF_TIME | nat.IN_COLLISION)) != 0
    assert np.array_equal(done, (f & nat.DONE) != 0)
    # o->done holds the values of the step just taken: for an env that auto-reset in that step they are the terminal
    # step's (all done) while the state already belongs to the new episode; everywhere else buffer and state agree
    fresh = (g.state["episode_step"].cpu().numpy() == 0)
    assert np.array_equal(done.astype(np.uint8)[~fresh], g.done.cpu().numpy()[~fresh])
    assert g.done.cpu().numpy()[fresh].all() and g.game_over.cpu().numpy()[fresh].all()
    assert not done[fresh].any()
    obs = g.obs.cpu().numpy()
    assert np.all(obs[..., 1] == N - 1)                      # everyone observes all 9 others
    oa = obs[..., 6:].reshape(E, N, N - 1, 7)
    key = np.rint(oa[..., 6].astype(np.float64) * 100)       # rows ascending in the rounded distance bucket
    assert np.all(np.diff(key, axis=-1) >= -1)               # (-1: float32 output may straddle a bucket edge)
    assert np.allclose(oa[..., 5], obs[..., 5:6] + oa[..., 4], atol=1e-6)   # combined radius
    d = np.hypot(oa[..., 0], oa[..., 1]) - oa[..., 5]
    assert np.allclose(d, oa[..., 6], atol=1e-4)             # |rel pos| - radii == dist_2_other
    rew = g.rewards.cpu().numpy()
    assert rew.min() >= -0.25 - 1e-6 and rew.max() <= 1.0 + 1e-6
    st = g.episode_stats().cpu().numpy()
    assert st[0] > 0 and st[0] == st[1] + st[2] + st[3]
    assert np.isfinite(g.state["pos_x"].cpu().numpy()).all()


# ---------------------------------------------------------------- GA3C-CADRL network (config 3 row; fp32 matrix cores)
def _ga3c_obs(rng, E, N, K):
    """plausible observation rows: num_other in [0, K], the first num_other slots filled, the rest zero"""
    obs = np.zeros((E, N, 6 + 7 * K), np.float32)
    num = rng.integers(0, K + 1, size=(E, N))
    obs[..., 1] = num
    obs[..., 2] = rng.uniform(0.1, 12.0, (E, N))
    obs[..., 3] = rng.uniform(-np.pi, np.pi, (E, N))
    obs[..., 4] = rng.uniform(0.5, 1.5, (E, N))
    obs[..., 5] = rng.uniform(0.2, 0.8, (E, N))
    oth = np.stack([rng.uniform(-8, 8, (E, N, K)), rng.uniform(-8, 8, (E, N, K)), rng.uniform(-1.5, 1.5, (E, N, K)),
                    rng.uniform(-1.5, 1.5, (E, N, K)), rng.uniform(0.2, 0.8, (E, N, K)), rng.uniform(0.4, 1.6, (E, N, K)),
                    rng.uniform(0.0, 10.0, (E, N, K))], axis=-1).astype(np.float32)
    oth *= (np.arange(K)[None, None, :] < num[..., None])[..., None]
    obs[..., 6:] = oth.reshape(E, N, 7 * K)
    return obs


@pytest.mark.parametrize("E,N,K", [(37, 5, 19), (3, 20, 19), (64, 4, 3), (9, 7, 25), (1, 1, 0)])
def test_ga3c_logits_and_actions_vs_numpy_network(E, N, K):
    """cagpu_ga3c against the numpy restatement of the TF graph on random observation rows: logits within fp32
    round-off, the same argmax wherever the two best logits are not within round-off of each other; agents that are
    done or run another policy are left alone"""
    nat, core, orc = _mods()
    from oracle.ga3c_ref import GA3CNet
    rng = np.random.default_rng(E * 1000 + N * 10 + K)
    g = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=1))
    pol = np.full((E, N), nat.POL_GA3C_CADRL)
    other = rng.random((E, N)) < 0.15
    pol[other] = nat.POL_RVO
    g.set_plugins(pol)
    done = (rng.random((E, N)) < 0.15) & ~other
    g.state["flags"] |= torch.from_numpy(np.where(done, nat.DONE, 0).astype(np.int32)).to(g.device)
    obs = _ga3c_obs(rng, E, N, K)
    g.obs.copy_(torch.from_numpy(obs))
    g.load_ga3c(keep_logits=True)
    g.ga3c_logits.fill_(-777.0)
    ext = torch.full((E, N, 2), -7.0, dtype=torch.float64, device=g.device)
    g.ga3c(ext)
    torch.cuda.synchronize()
    live = ~other & ~done
    net = GA3CNet()
    want = net.logits(net.policy_vector(obs.reshape(E * N, -1))).reshape(E, N, 11)
    got = g.ga3c_logits.cpu().numpy()
    ex = ext.cpu().numpy()
    assert np.all(got[~live] == -777.0) and np.all(ex[~live] == -7.0)
    if live.any():
        np.testing.assert_allclose(got[live], want[live], rtol=1e-4, atol=2e-4)
        srt = np.sort(want[live], axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-3
        assert clear.mean() > 0.9
        assert np.array_equal(ex[live][clear, 0], np.argmax(want[live], axis=1)[clear])
        assert np.all(ex[live][:, 1] == 0.0) and np.all((ex[live][:, 0] >= 0) & (ex[live][:, 0] <= 10))
        assert np.array_equal(ex[live][:, 0], np.argmax(got[live], axis=1))      # first maximum, like np.argmax
    # the packed list holds exactly the live rows; without it (CaNet.rows_scratch = NULL: whole tiles) the same bits
    assert g.ga3c_rows() == int(live.sum())
    # the scratch needs no initialisation: garbage in it (list, count and the two tagged counters) changes nothing
    sc = g._net_tensors["rows_scratch"]
    sc.copy_(torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, size=sc.numel(), dtype=np.int64).astype(np.int32)))
    g.ga3c_logits.fill_(-777.0)
    ext1 = torch.full((E, N, 2), -7.0, dtype=torch.float64, device=g.device)
    g.ga3c(ext1)
    torch.cuda.synchronize()
    assert g.ga3c_rows() == int(live.sum())
    assert np.array_equal(g.ga3c_logits.cpu().numpy(), got) and np.array_equal(ext1.cpu().numpy(), ex)
    g._net.rows_scratch = None
    g.ga3c_logits.fill_(-777.0)
    ext2 = torch.full((E, N, 2), -7.0, dtype=torch.float64, device=g.device)
    g.ga3c(ext2)
    torch.cuda.synchronize()
    assert np.array_equal(g.ga3c_logits.cpu().numpy(), got) and np.array_equal(ext2.cpu().numpy(), ex)


@pytest.mark.parametrize("live_rows", [16000, 20000, 24500, 27000, 30000, 32768, 33000, 50000, 81920])
def test_ga3c_balanced_tile_heights_cover_every_row(live_rows):
    """ga3c_kernel spreads the live rows' 16-row blocks evenly over the tiles of its rounds (64 / 48 / 32-row tiles in one
    launch, csrc/cagpu_ga3c.inc tile_of): at row counts on every branch of that plan -- uniform 32-row tiles, lo = 2 / 3
    with and without a remainder, exactly one full round, two and three rounds -- every live row is evaluated, no other row
    is touched, the logits are those of the numpy network and bit for bit those of the unpacked launch (another plan)"""
    nat, core, orc = _mods()
    from oracle.ga3c_ref import GA3CNet
    E, N, K = 4096, 20, 19
    rng = np.random.default_rng(live_rows)
    g = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=1))
    g.set_plugins(nat.POL_GA3C_CADRL)
    live = np.zeros(E * N, bool)
    live[rng.permutation(E * N)[:live_rows]] = True
    live = live.reshape(E, N)
    g.state["flags"] |= torch.from_numpy(np.where(live, 0, nat.DONE).astype(np.int32)).to(g.device)
    obs = _ga3c_obs(rng, E, N, K)
    g.obs.copy_(torch.from_numpy(obs))
    g.load_ga3c(keep_logits=True)
    g.ga3c_logits.fill_(-777.0)
    ext = torch.full((E, N, 2), -7.0, dtype=torch.float64, device=g.device)
    g.ga3c(ext)
    torch.cuda.synchronize()
    assert g.ga3c_rows() == live_rows
    got = g.ga3c_logits.cpu().numpy()
    ex = ext.cpu().numpy()
    assert np.all(got[~live] == -777.0) and np.all(ex[~live] == -7.0)
    assert np.all(got[live] != -777.0) and np.all(ex[live][:, 1] == 0.0)
    assert np.array_equal(ex[live][:, 0], np.argmax(got[live], axis=1))
    pick = np.flatnonzero(live.reshape(-1))
    pick = pick[rng.permutation(pick.size)[:1500]]
    net = GA3CNet()
    want = net.logits(net.policy_vector(obs.reshape(E * N, -1)[pick]))
    np.testing.assert_allclose(got.reshape(E * N, 11)[pick], want, rtol=1e-4, atol=2e-4)
    g._net.rows_scratch = None
    g.ga3c_logits.fill_(-777.0)
    ext2 = torch.full((E, N, 2), -7.0, dtype=torch.float64, device=g.device)
    g.ga3c(ext2)
    torch.cuda.synchronize()
    assert np.array_equal(g.ga3c_logits.cpu().numpy(), got) and np.array_equal(ext2.cpu().numpy(), ex)


def _ga3c_logits_f64(w, x):
    """the graph of oracle/ga3c_ref.GA3CNet.logits evaluated in float64 (normalisation in float32 like the graph: it is
    the network's INPUT): the yardstick for what a float32 evaluation -- numpy's or the kernel's -- loses"""
    x = np.asarray(x, dtype=np.float32)
    B = x.shape[0]
    seq = x[:, 0].astype(np.int32)
    xn = ((x - w["input_mean"]) / w["input_std"]).astype(np.float32).astype(np.float64)
    host, others = xn[:, 1:5], xn[:, 5:].reshape(B, 19, 7)
    W = {k: v.astype(np.float64) for k, v in w.items()}
    sg = lambda v: 1.0 / (1.0 + np.exp(-v))
    h, c = np.zeros((B, 64)), np.zeros((B, 64))
    for t in range(19):
        z = np.concatenate([others[:, t], h], axis=1) @ W["lstm_kernel"] + W["lstm_bias"]
        i, j, f, o = np.split(z, 4, axis=1)
        cn = sg(f + 1.0) * c + sg(i) * np.tanh(j)
        hn = sg(o) * np.tanh(cn)
        live = (t < seq)[:, None]
        c, h = np.where(live, cn, c), np.where(live, hn, h)
    a = np.concatenate([host, h], axis=1)
    for n in ("layer1", "layer2", "fc1"):
        a = np.maximum(a @ W[n + "_kernel"] + W[n + "_bias"], 0)
    return a @ W["logits_p_kernel"] + W["logits_p_bias"]


def test_ga3c_split_operand_network_is_float32_accurate():
    """The kernel multiplies float32 operands as two fp16 planes each (three of the four plane products; rounds 2 - 4: three
    bf16 planes, six of nine): its logits must sit as close to a float64 evaluation of the graph as numpy's float32 evaluation
    does (same order of magnitude: both are float32-accumulated), far inside the parity bar -- i.e. the split is a
    float32-class product, not an fp16 one"""
    nat, core, orc = _mods()
    from oracle.ga3c_ref import GA3CNet
    E, N, K = 256, 16, 19
    rng = np.random.default_rng(77)
    g = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=1))
    g.set_plugins(nat.POL_GA3C_CADRL)
    obs = _ga3c_obs(rng, E, N, K)
    g.obs.copy_(torch.from_numpy(obs))
    g.load_ga3c(keep_logits=True)
    g.ga3c(torch.zeros((E, N, 2), dtype=torch.float64, device=g.device))
    torch.cuda.synchronize()
    got = g.ga3c_logits.cpu().numpy().reshape(E * N, 11).astype(np.float64)
    net = GA3CNet()
    x = net.policy_vector(obs.reshape(E * N, -1))
    ref64 = _ga3c_logits_f64(net.w, x)
    ref32 = net.logits(x).astype(np.float64)
    err_gpu, err_np = np.abs(got - ref64), np.abs(ref32 - ref64)
    scale = np.abs(ref64).max()
    print("logits up to %.2f: |gpu - f64| max %.3g mean %.3g; |numpy f32 - f64| max %.3g mean %.3g" % (
        scale, err_gpu.max(), err_gpu.mean(), err_np.max(), err_np.mean()))
    # measured: logits up to 64; |gpu - f64| max 3.9e-5 mean 1.41e-6 (three bf16 planes, six products: 3.2e-5 / 1.66e-6);
    # |numpy f32 - f64| max 2.0e-5 mean 1.33e-6
    assert err_gpu.max() < 3 * err_np.max() and err_gpu.mean() < 2 * err_np.mean()
    # (an fp16-precision product would be off by ~1e-2 here: three orders of magnitude above the bound)


def test_ga3c_pack_is_the_two_plane_split_and_required():
    """cagpu_ga3c_pack: every packed weight is its two fp16 planes hi = fp16(w), lo = fp16(w - hi) (round to nearest, fp16
    denormals kept; fragment order of csrc/cagpu_ga3c.inc), hi + lo within 2^-22 of w, rows past a matrix's K are zero; the
    LSTM kernel's columns carry the gates' 2^z scale (the float32 product with -log2 e, the j gate's with -2 log2 e:
    csrc/cagpu_ga3c.inc, struct Gate); cagpu_ga3c refuses a CaNet without it"""
    nat, core, orc = _mods()
    g = core.BatchedSim(core.make_params(4, 3, max_obs=19, sort_mode=1))
    g.set_plugins(nat.POL_GA3C_CADRL)
    g.load_ga3c()
    torch.cuda.synchronize()
    ts = g._net_tensors
    pk = ts["packed"].cpu().numpy().view(np.float16).reshape(-1, 2, 64, 8)     # [(kb, cb), plane, lane, e]
    assert pk.shape[0] * 2 * 64 * 16 == int(g.lib.cagpu_ga3c_packed_bytes())
    planes = pk.astype(np.float32)                                              # fp16 -> float32, exact
    at = 0
    for name, row0, nkb in (("lstm_kernel", 7, 2), ("layer1_kernel", 4, 2), ("layer2_kernel", 0, 8), ("fc1_kernel", 0, 8)):
        w = ts[name].cpu().numpy()
        blk = planes[at:at + nkb * 16].reshape(nkb, 16, 2, 64, 8)
        at += nkb * 16
        lane = np.arange(64)
        m, q = lane & 15, lane >> 4
        for kb in range(nkb):
            for cb in range(16):
                k = row0 + kb * 32 + 8 * q[:, None] + np.arange(8)[None, :]    # [lane, e]
                want = w[k, (cb * 16 + m)[:, None]]
                if name == "lstm_kernel":
                    want = want * np.float32(-2.88539008177792681 if cb // 4 == 1 else -1.44269504088896341)
                p2 = blk[kb, cb]
                hi = want.astype(np.float16).astype(np.float32)
                lo = (want - hi).astype(np.float16).astype(np.float32)
                assert np.array_equal(p2[0], hi) and np.array_equal(p2[1], lo), (name, kb, cb)
                assert np.all(np.abs((p2[0].astype(np.float64) + p2[1]) - want) <= np.abs(want) * 2.0 ** -22 + 3e-8)
    assert at == planes.shape[0]
    g._net.packed = None
    with pytest.raises(nat.CagpuError, match="packed"):
        g.ga3c(torch.zeros((4, 3, 2), dtype=torch.float64, device=g.device))
    bad = nat.CaNet()
    assert g.lib.cagpu_ga3c_pack(ctypes.byref(bad), ts["packed"].data_ptr(), ts["packed"].numel(), None) == nat.CA_EINVAL
    assert g.lib.cagpu_ga3c_pack(ctypes.byref(g._nets[0][0]), ts["packed"].data_ptr(), 16, None) == nat.CA_EINVAL


def test_ga3c_with_nothing_alive_touches_nothing():
    """every GA3C-CADRL agent done (or none in the batch): the packed list is empty, every workgroup of the network launch
    leaves at once, ext_actions / logits keep what they held"""
    nat, core, orc = _mods()
    for pol in (nat.POL_GA3C_CADRL, nat.POL_RVO):
        g = core.BatchedSim(core.make_params(300, 7, max_obs=19, sort_mode=1))
        g.set_plugins(pol)
        g.state["flags"] |= nat.DONE
        g.load_ga3c(keep_logits=True)
        g.ga3c_logits.fill_(-777.0)
        ext = torch.full((300, 7, 2), -7.0, dtype=torch.float64, device=g.device)
        g.ga3c(ext)
        torch.cuda.synchronize()
        assert g.ga3c_rows() == 0
        assert bool((ext == -7.0).all()) and bool((g.ga3c_logits == -777.0).all())


def test_ga3c_needs_loaded_network():
    nat, core, orc = _mods()
    g = core.BatchedSim(core.make_params(2, 3, max_obs=19))
    g.set_plugins(nat.POL_GA3C_CADRL)
    with pytest.raises(nat.CagpuError):
        g.step()


@pytest.mark.parametrize("N,mixed", [(6, False), (10, True)])
def test_ga3c_step_vs_oracle_reinjected(N, mixed):
    """GA3C-CADRL agents stepping: every step starts from the oracle's state AND observation, the network's choice is
    compared agent by agent (ties within round-off excepted) and the resulting states wherever the choices agree"""
    nat, core, orc = _mods()
    from gym_collision_avoidance_amd.envs import test_cases as tc
    E, T = 48, 40
    table = tc.fixture_table(N).astype(np.float32).astype(np.float64)
    o, g = _pair(E, N, K=19, sort_mode=1)
    pol = np.full((E, N), orc.POL_GA3C_CADRL)
    if mixed:
        pol[:, 1::3] = orc.POL_RVO
        pol[:, 2::5] = orc.POL_NONCOOP
    o.set_policies(pol)
    g.set_plugins(pol)
    g.load_ga3c(keep_logits=True)
    o.reset(table[:E])
    agree_total, n_total = 0, 0
    for t in range(T):
        _upload(o, g)
        g.obs.copy_(torch.from_numpy(o.obs.astype(np.float32)))
        o.step()
        g.step()
        torch.cuda.synchronize()
        q = o.ga3c_index.reshape(E, N)
        asked = q >= 0
        gi = g._ga3c_ext.cpu().numpy()[..., 0]
        same = (gi == q) | ~asked
        lg = np.sort(g.ga3c_logits.cpu().numpy(), axis=-1)
        assert np.all((lg[..., -1] - lg[..., -2])[~same] < 1e-3), "step %d: a clear-cut choice differs" % t
        agree_total += int((same & asked).sum())
        n_total += int(asked.sum())
        ok = same.all(axis=1)
        for n in ("pos_x", "pos_y", "heading", "vel_x", "vel_y"):
            np.testing.assert_allclose(g.state[n].cpu().numpy()[ok], o.view(n)[ok], rtol=0, atol=TOL, err_msg="%s @%d" % (n, t))
        gf = g.state["flags"].cpu().numpy().astype(np.uint32)
        assert np.array_equal(gf[ok] & MASK, o.view("flags")[ok] & MASK)
        np.testing.assert_allclose(g.obs.cpu().numpy()[ok], o.obs[ok], rtol=0, atol=TOL)
    assert n_total > 1000 and agree_total >= 0.995 * n_total, (agree_total, n_total)


def test_ga3c_full_size_config3():
    """BASELINE config 3 geometry: 4096 envs x 20 GA3C-CADRL agents (K = 19, closest_last): a circle swap with
    per-env jitter; every choice is a valid action, most agents make progress, repeated runs are deterministic"""
    nat, core, orc = _mods()
    from gym_collision_avoidance_amd.envs import test_cases as tc
    E, N = 4096, 20
    rng = np.random.default_rng(3)
    base = tc.gen_circle_test_case(N, 10.0)
    cases = np.repeat(base[None], E, axis=0)
    cases[..., :4] += rng.uniform(-0.4, 0.4, (E, N, 4))
    cases[..., 4] = rng.uniform(0.6, 1.4, (E, N))
    cases[..., 5] = rng.uniform(0.2, 0.5, (E, N))

    def run():
        g = core.BatchedSim(core.make_params(E, N, max_obs=19, sort_mode=1))
        g.set_plugins(nat.POL_GA3C_CADRL)
        g.load_ga3c()
        g.reset(cases)
        for _ in range(30):
            g.step()
        torch.cuda.synchronize()
        return g
    a, b = run(), run()
    idx = a._ga3c_ext.cpu().numpy()[..., 0]
    assert np.all((idx >= 0) & (idx <= 10) & (idx == np.round(idx)))
    for n in ("pos_x", "pos_y", "heading"):
        assert torch.equal(a.state[n], b.state[n])
    d0 = np.hypot(cases[..., 0] - cases[..., 2], cases[..., 1] - cases[..., 3])
    d1 = np.hypot(a.state["pos_x"].cpu().numpy() - cases[..., 2], a.state["pos_y"].cpu().numpy() - cases[..., 3])
    assert (d1 < d0 - 1.0).mean() > 0.9
    assert np.isfinite(a.obs.cpu().numpy()).all()


@pytest.mark.parametrize("seed", range(8))
def test_fuzzed_config_constants_vs_oracle(seed):
    """every constant of CaParams drawn at random (finite sensing horizon, fewer ORCA neighbours than agents, clipped
    observations, all sort / game-over modes, wiggle and time-step rewards, DT, thresholds, RVO horizon / collaboration),
    RVO + non-cooperative agents from the fixture tables, re-injected each step"""
    nat, core, orc = _mods()
    from gym_collision_avoidance_amd.envs import test_cases as tc
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([3, 5, 6, 8, 10]))
    E = int(rng.integers(20, 120))
    K = int(rng.integers(1, N + 3))
    o, g = _pair(E, N, K, sort_mode=int(rng.integers(0, 3)), game_over_mode=int(rng.integers(0, 3)),
                 max_time_ratio=float(rng.uniform(1.5, 8.0)), dt=float(rng.choice([0.1, 0.2, 0.05])),
                 near_goal=float(rng.uniform(0.1, 0.5)), getting_close=float(rng.uniform(0.1, 0.6)),
                 rvo_max_neighbors=int(rng.integers(1, N + 1)), obs_clip=int(rng.integers(0, K + 1)))
    extra = dict(sensing_horizon=float(rng.choice([np.inf, 4.0, 7.5])), reward_time_step=float(rng.choice([0.0, -0.01])),
                 reward_wiggly=float(rng.choice([0.0, -0.02])), wiggly_threshold=float(rng.choice([np.inf, 0.2])),
                 reward_collision=float(rng.uniform(-0.5, -0.1)), reward_at_goal=float(rng.uniform(0.5, 2.0)),
                 rvo_time_horizon=float(rng.uniform(2.0, 8.0)), rvo_collab_coeff=float(rng.uniform(0.2, 0.8)),
                 max_heading_change=float(rng.uniform(0.6, 1.2)))
    for p in (o.p, g.p):
        for k_, v in extra.items():
            setattr(p, k_, v)
        vals = [p.reward_at_goal, p.reward_collision, p.reward_time_step, p.reward_wiggly]
        p.reward_min, p.reward_max = min(vals), max(vals)
    pol = np.where(rng.random((E, N)) < 0.8, orc.POL_RVO, orc.POL_NONCOOP).astype(np.int32)
    dyn = np.where(rng.random((E, N)) < 0.8, orc.DYN_UNICYCLE, orc.DYN_MAX_TURN_RATE).astype(np.int32)
    o.s["policy"][:] = pol.reshape(-1)
    o.s["dynamics"][:] = dyn.reshape(-1)
    g.set_plugins(pol, dyn)
    table = tc.fixture_table(N).astype(np.float32).astype(np.float64)
    cases = table[rng.integers(0, 500, E)]
    o.reset(cases)
    g.reset(cases)
    _compare_reset(o, g)
    for t in range(50):
        _upload(o, g)
        o.step()
        g.step()
        _compare(o, g, tol=2e-5 if np.isfinite(extra["wiggly_threshold"]) else TOL, what="seed %d step %d" % (seed, t))


def test_auto_reset_without_precomputed_observations():
    """CaAutoReset.reset_obs == NULL (a caller of the C ABI that did not precompute the reset observations): the kernel
    re-runs the sensing phases for the envs that reset; outputs and state are bit-identical to the table path"""
    nat, core, orc = _mods()
    N, E = 10, 300
    table = gu.fixtures(N)
    sims = []
    for with_obs in (True, False):
        g = core.BatchedSim(core.make_params(E, N))
        g.set_plugins(nat.POL_RVO)
        g.set_fixture_table(table)
        if not with_obs:
            g._ar.reset_obs = None
        g.reset_from_table()
        sims.append(g)
    for t in range(260):
        for g in sims:
            g.step()
        if t % 20 == 19:
            assert torch.equal(sims[0].obs, sims[1].obs), t
    assert float(sims[0].episode_stats()[0]) > 100        # plenty of auto-resets happened
    for n in F64 + ("flags", "step_num", "episode_step", "reset_count", "env_stats"):
        x, y = sims[0].state[n], sims[1].state[n]
        if n == "flags":   # (PLAN_VALID is bookkeeping: without reset observations the unpipelined kernel runs)
            x, y = x & ~nat.PLAN_VALID, y & ~nat.PLAN_VALID
        assert torch.equal(x, y), n
    assert torch.equal(sims[0].rewards, sims[1].rewards) and torch.equal(sims[0].done, sims[1].done)


@pytest.mark.parametrize("N,side,seed", [(2, 4.0, 1), (4, 5.0, 2), (10, 4.0, 3), (10, (4.0, 8.0), 4), (20, 8.0, 5)])
def test_device_scenarios_match_host_generator(N, side, seed):
    """cagpu_generate_cases (section 8 f3: get_testcase_random on the device) against the host generator -- itself
    bit-identical to the reference under np.random -- driven by the same Philox stream: the same draw order and the same
    accept / reject decisions give the same scenarios (cos / sin of the circle families differ by an ulp between the
    device's and the host's libm, hence 1e-9 and a little room for a borderline decision)"""
    nat, core, orc = _mods()
    from tests.test_host_logic import host_cases_from_philox
    C_ = 96
    g = core.BatchedSim(core.make_params(4, N))
    got, status = g.generate_cases(C_, seed, side_length=side, return_status=True)
    torch.cuda.synchronize()
    got, status = got.cpu().numpy(), status.cpu().numpy()
    want, kinds = host_cases_from_philox(seed, C_, N, side)
    assert set(kinds) == {"swap", "circle", "rand"} and not status.any()
    same = np.abs(got - want).reshape(C_, -1).max(axis=1) <= 1e-9
    assert same.mean() >= 0.97, (same.mean(), [k for k, s_ in zip(kinds, same) if not s_])
    # a case is a function of (seed, case index) alone
    again = g.generate_cases(17, seed, side_length=side).cpu().numpy()
    assert np.array_equal(again, got[:17])
    other = g.generate_cases(17, seed + 1, side_length=side).cpu().numpy()
    assert not np.array_equal(other, got[:17])


@pytest.mark.parametrize("N,E,pipeline,ragged", [(10, 4096, True, 0), (10, 3073, True, 0), (10, 300, False, 0), (4, 333, True, 0),
                                                  (4, 200, True, 1), (20, 130, True, 0), (50, 24, True, 0)])
def test_step_rewrites_every_output_element(N, E, pipeline, ragged):
    """Every step launch writes EVERY element of obs, rewards, done and game_over (done agents, empty slots of a ragged
    batch, envs that auto-reset in the step, the last tile of a batch included): the outputs carry nothing from one step
    to the next, which is what lets BatchedSim.fresh_outputs hand each step newly allocated tensors instead of copies"""
    nat, core, orc = _mods()
    from tests import golden_util as gu
    table = gu.suite_cases("ragged4") if ragged else np.load(
        os.path.join(REPO, "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n%d" % N]
    C_ = len(table)
    g = core.BatchedSim(core.make_params(E, N, ragged=ragged), pipeline=pipeline)
    g.set_plugins(nat.POL_RVO)
    g.reset(table[np.arange(E) % C_])
    g.set_fixture_table(table)
    poisoned = 0
    for t in range(260):
        if t % 20 == 19 or t > 250:  # (also steps in which envs auto-reset: a 4-agent case ends after ~60 steps)
            g.obs.fill_(float("nan")); g.rewards.fill_(float("nan")); g.done.fill_(0xFF); g.game_over.fill_(0xFF)
            poisoned += 1
        g.step()
        if t % 20 == 19 or t > 250:
            torch.cuda.synchronize()
            assert not torch.isnan(g.obs).any(), t
            assert not torch.isnan(g.rewards).any(), t
            assert int(g.done.max()) <= 1 and int(g.game_over.max()) <= 1, t
    assert poisoned > 15 and (N > 20 or int(g.state["reset_count"].max()) >= 1)  # (a 50-agent crowd takes longer)
    # the same through fresh_outputs: new tensors every step, the old ones keep their contents, the state advances the same
    h = core.BatchedSim(core.make_params(E, N, ragged=ragged), pipeline=pipeline)
    h.set_plugins(nat.POL_RVO)
    h.reset(table[np.arange(E) % C_])
    h.set_fixture_table(table)
    h.fresh_outputs = True
    kept = []
    for t in range(260):
        o_, r_, go_ = h.step()
        if t in (3, 100, 259):
            kept.append((o_, o_.clone(), r_, r_.clone()))
    torch.cuda.synchronize()
    assert len({k_[0].data_ptr() for k_ in kept}) == 3
    for o_, oc, r_, rc in kept:
        assert torch.equal(o_, oc) and torch.equal(r_, rc)
    assert torch.equal(h.obs, g.obs) and torch.equal(h.rewards, g.rewards) and torch.equal(h.done, g.done)
    for n in F64:
        assert torch.equal(h.state[n], g.state[n]), n


def test_device_ragged_scenarios_match_host_generator():
    """cagpu_generate_cases_ragged (get_testcase_random with num_agents=None and the reference's default side_length
    list, test_cases.py:224-241 / config.py:118-131): every case draws its agent count, then its side length from the
    range that holds the count; against the host generator under the same Philox stream, rows past the count zero"""
    nat, core, orc = _mods()
    from tests.test_host_logic import host_cases_from_philox
    C_, N = 128, 10
    sides = [{"num_agents": [0, 5], "side_length": [4, 5]}, {"num_agents": [5, np.inf], "side_length": [6, 8]}]
    dev_sides = [{"num_agents": [0, 5], "side_length": [4, 5]}, {"num_agents": [5, 1 << 20], "side_length": [6, 8]}]
    g = core.BatchedSim(core.make_params(4, N, ragged=1))
    got, status, counts = g.generate_cases(C_, 77, side_length=dev_sides, num_agents=(2, N), return_status=True,
                                           return_counts=True)
    torch.cuda.synchronize()
    got, status, counts = got.cpu().numpy(), status.cpu().numpy(), counts.cpu().numpy()
    want, kinds = host_cases_from_philox(77, C_, N, sides, num_agents=(2, N))
    assert not status.any() and set(kinds) == {"swap", "circle", "rand"}
    assert np.array_equal(counts, (want[..., 5] > 0).sum(1)) and set(counts) == set(range(2, N + 1))
    assert np.array_equal(got[..., 5] > 0, want[..., 5] > 0) and not got[got[..., 5] <= 0].any()
    same = np.abs(got - want).reshape(C_, -1).max(axis=1) <= 1e-9
    assert same.mean() >= 0.97, (same.mean(), [k for k, s_ in zip(kinds, same) if not s_])
    # as the on-device auto-reset table of a ragged batch: absent slots stay absent through resets, nothing blows up
    sim = core.BatchedSim(core.make_params(256, N, ragged=1))
    table = g.generate_cases(512, 78, side_length=dev_sides, num_agents=(2, N))
    sim.set_plugins(nat.POL_RVO)
    sim.reset(table[:256])
    sim.set_fixture_table(table)
    for _ in range(300):
        sim.step()
    torch.cuda.synchronize()
    fl = sim.state["flags"].cpu().numpy().astype(np.uint32)
    nres = sim.state["reset_count"].cpu().numpy()
    tab = table.cpu().numpy()
    assert np.isfinite(sim.state["pos_x"].cpu().numpy()).all() and nres.max() >= 1
    case = (np.arange(256) + nres.astype(np.int64) * 256) % 512
    assert np.array_equal((fl >> 16 & 1) == 1, tab[case][..., 5] <= 0)
    # bad arguments are refused: a count range outside the side ranges, a count above the table width
    with pytest.raises(nat.CagpuError):
        g.generate_cases(4, 1, side_length=[{"num_agents": [0, 5], "side_length": [4, 5]}], num_agents=(2, N))
    with pytest.raises(nat.CagpuError):
        g.generate_cases(4, 1, side_length=4.0, num_agents=(2, N + 1))


def test_device_scenarios_statistics_and_training_reset():
    """4096 generated 10-agent scenarios: family mix 15 / 15 / 70 %, the reference's clearance and trip-length rules,
    speed = max of two uniforms, uniform radii; then used as the on-device auto-reset table of a stepping batch"""
    nat, core, orc = _mods()
    N, C_ = 10, 4096
    g = core.BatchedSim(core.make_params(256, N))
    cases, status = g.generate_cases(C_, 2024, side_length=(4.0, 6.0), return_status=True)
    torch.cuda.synchronize()
    cs, status = cases.cpu().numpy(), status.cpu().numpy()
    assert not status.any() and np.isfinite(cs).all()
    start, goal, speed, rad = cs[..., 0:2], cs[..., 2:4], cs[..., 4], cs[..., 5]
    swap = (start[:, 0, 1] == 0.0) & (goal[:, 0, 1] == 0.0) & (start[:, 0, 0] == -start[:, 1, 0])
    circle = ~swap & (np.abs(start + goal).reshape(C_, -1).max(axis=1) < 1e-9)   # antipodal around the origin
    frac = np.array([swap.mean(), circle.mean(), (~swap & ~circle).mean()])
    assert np.all(np.abs(frac - [0.15, 0.15, 0.70]) < 0.03), frac
    d_start = np.linalg.norm(start[:, :, None] - start[:, None], axis=-1)
    d_goal = np.linalg.norm(goal[:, :, None] - goal[:, None], axis=-1)
    clr = rad[:, :, None] + rad[:, None] + 0.2
    off = ~np.eye(N, dtype=bool)[None]
    assert np.all((d_start >= clr) | ~off) and np.all((d_goal >= clr) | ~off)
    trip = np.linalg.norm(start - goal, axis=-1)
    assert np.all(trip[~swap & ~circle] > 0.5 * 4.0)      # > half of a side that only grows from its first value
    assert np.all((speed >= 0.5) & (speed <= 2.0) & (rad >= 0.2) & (rad <= 0.8))
    assert abs(speed.mean() - 1.5) < 0.01 and abs(rad.mean() - 0.5) < 0.01   # E max(U1, U2) = lo + 2/3 (hi - lo)
    # training-mode resets straight from the generated table: nothing touches the host
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(cases)
    g.reset_from_table()
    g.rollout(400)
    torch.cuda.synchronize()
    st = g.episode_stats().cpu().numpy()
    assert st[0] > 100 and np.isfinite(g.obs.cpu().numpy()).all()


def test_random_reset_headings_training_mode():
    """CaAutoReset.heading_seed (test_cases.py:558-559, training mode): auto-resets draw the initial heading uniformly
    in [-pi, pi) on the device; the observation handed back is that of the reset state (cagpu_observe reproduces it);
    a heading is a function of (seed, global env id, reset count, agent) alone -- shards see what one big batch sees"""
    nat, core, orc = _mods()
    N, E = 10, 256
    table = np.load(os.path.join(REPO, "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n10"]

    def run(E_, off, seed, steps=260):
        g = core.BatchedSim(core.make_params(E_, N))
        g.set_plugins(nat.POL_RVO)
        g.set_fixture_table(table, env_id_offset=off, case_stride=E, heading_seed=seed)
        g.reset_from_table()
        first = {}
        prev = g.state["reset_count"].cpu().numpy().copy()
        for t in range(steps):
            g.step()
            rc = g.state["reset_count"].cpu().numpy()
            new = np.nonzero(rc != prev)[0]
            if len(new):
                hd = g.state["heading"].cpu().numpy()
                obs = g.obs.cpu().numpy().copy()
                ob2 = g.observe().cpu().numpy() if hasattr(g, "observe") else None
                for e in new:
                    first[(off + int(e), int(rc[e]))] = hd[e].copy()
                if ob2 is not None:
                    assert np.array_equal(ob2[new], obs[new])      # the reset observation IS the observation of the reset state
            prev = rc.copy()
        return first
    a = run(E, 0, 7)
    hs = np.array(list(a.values()))
    assert len(a) > 200 and np.all((hs >= -np.pi) & (hs < np.pi))
    assert abs(hs.mean()) < 0.15 and abs(hs.std() - np.pi / np.sqrt(3)) < 0.15           # uniform on [-pi, pi)
    assert len(np.unique(np.round(hs.ravel(), 12))) > 0.99 * hs.size
    from oracle.philox_ref import philox4x32_10                                             # the exact values
    for (ge, rc), hd in list(a.items())[:40]:
        for ag in range(N):
            w = philox4x32_10((ge & 0xFFFFFFFF, ge >> 32, rc, ag), (7, 0))
            u = ((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) / 9007199254740992.0
            assert hd[ag] == -np.pi + 2.0 * np.pi * u, (ge, rc, ag)
    b = run(E // 2, E // 2, 7)                                                              # the upper half as its own shard
    common = [k for k in b if k in a]
    assert len(common) > 50 and all(np.array_equal(a[k], b[k]) for k in common)
    c = run(E, 0, 8, steps=200)
    assert any(not np.array_equal(a[k], c[k]) for k in c if k in a)
    goal = run(E, 0, 0, steps=200)                                                          # seed 0: towards the goal
    assert len(goal) > 50


# ---------------------------------------------------------------- round 3: the ORCA phases of the STEP kernels bit for bit
def _state_bits(g):
    out = {n: g.state[n].cpu().numpy().copy() for n in F64 + ("last_action", "step_num", "episode_step", "reset_count",
                                                              "env_stats")}
    out["flags"] = g.state["flags"].cpu().numpy() & ~(1 << 17)           # (PLAN_VALID is bookkeeping, not state)
    for n in ("obs", "rewards", "done", "game_over", "actions", "orca_vel"):
        out[n] = getattr(g, n).cpu().numpy().copy()
    return out


def _assert_same_bits(a, b, what):
    for n in a:
        x, y = a[n], b[n]
        same = np.array_equal(x.view(np.uint8) if x.dtype.kind == "f" else x, y.view(np.uint8) if y.dtype.kind == "f" else y)
        assert same, "%s differs (%s): %d of %d elements" % (n, what, (x != y).sum(), x.size)


@pytest.mark.parametrize("N,E,steps,pipeline", [(10, 3073, 12, True), (10, 4096, 12, True), (10, 4096, 12, False),
                                                (10, 1024, 12, True), (10, 1601, 12, True), (10, 2900, 12, True),
                                                (10, 5200, 8, True), (10, 8200, 6, True), (20, 300, 12, True), (50, 40, 10, True),
                                                (4, 333, 12, True), (4, 333, 12, False), (2, 500, 10, True), (3, 211, 10, True),
                                                (5, 130, 10, True), (6, 97, 10, True), (8, 250, 10, True)])
def test_step_kernel_orca_velocities_bit_exact(N, E, steps, pipeline):
    """CaOut.orca_vel -- the velocity the ORCA phases of the STEP kernel itself chose (branch-free half-planes, divq /
    sqrtq, parallel 1-D programmes + scan, lp3_wave8 / lp3_group / the wave-per-agent programme of N > 16; in the
    pipelined kernel the plan computed beside the previous step's sensing half) -- against the oracle's RVO2 restatement,
    bit for bit, on mid-episode states.  GPU and oracle continue from IDENTICAL bits each step (the GPU state is copied
    into the oracle), so every float input of ORCA is the same on both sides and the pipelined kernel stays on its fast
    path: the velocities compared at step t+1 were planned during launch t."""
    nat, core, orc = _mods()
    table = gu.fixtures(N)
    po, pg = orc.default_params(E, N), core.make_params(E, N)
    o, g = orc.Oracle(po), core.BatchedSim(pg, record_actions=True, pipeline=pipeline)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    cases = table[np.arange(E) % table.shape[0]]
    o.reset(cases)
    for _ in range(20 if N <= 3 else (60 if N <= 10 else 25)):   # mid-episode: most programmes have violated lines, some infeasible
        o.step()
    _upload(o, g)
    queried = 0
    for t in range(steps):
        _download(g, o)
        was_done = (o.s["flags"] & orc.DONE) != 0
        o.step()
        g.step()
        kern = nat.lib().cagpu_last_kernel().decode()
        if pipeline and (N in (2, 3, 4, 5, 6, 8) or (N == 10 and (E <= 4096 or E >= 8192))):
            assert kern.startswith("ca_pipe_kernel<%d, %d, false>" % (N, _pipe_tile(N, E))), kern
            if t > 0:   # the fast path: every agent that is queried next holds a valid plan
                assert (g.state["flags"].cpu().numpy().reshape(-1) >> 17 & 1).all()
        else:
            assert kern.startswith("ca_kernel<"), kern
        got, want = g.orca_vel.cpu().numpy().reshape(-1, 2), o.orca_vel.reshape(-1, 2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
            "step %d: %d of %d ORCA velocities differ (max %g)" % (t, (got != want).any(1).sum(), (~was_done).sum(),
                                                                   np.abs(got - want).max())
        assert not g.actions.cpu().numpy().reshape(-1, 2)[was_done].any()
        queried += int((~was_done).sum())
        _compare(o, g, what="N=%d step %d" % (N, t))
    assert queried > E * N * steps // 4


@pytest.mark.parametrize("E,chunks", [(100, [1] * 40 + [7, 1, 1, 30, 2, 150]), (4096, [1] * 6 + [5, 1, 40, 1, 1]),
                                      (1024, [1] * 6 + [5, 1, 40, 1, 1]), (2048, [1, 1, 9, 1, 30]), (3001, [1, 1, 9, 1, 30]),
                                      (6144, [1, 1, 6, 1, 25]),       # BASELINE configs[1], partial rounds, the 1.5-round grid
                                      (9001, [1, 1, 6, 1, 25])])      # 9001 envs: more workgroups than resident slots
def test_pipelined_equals_unpipelined_bit_for_bit(E, chunks):
    """CaState.next_action changes WHEN the RVO policy of a step is computed (beside the previous step's sensing half
    instead of at the start of the step), never what it computes: state and outputs of a free run with auto-resets --
    single steps and fused rollouts mixed, the plan crossing launch boundaries -- are identical bit for bit."""
    nat, core, orc = _mods()
    N = 10
    table = gu.fixtures(N)
    sims = []
    for pl in (True, False):
        g = core.BatchedSim(core.make_params(E, N), record_actions=True, pipeline=pl)
        g.set_plugins(nat.POL_RVO)
        g.set_fixture_table(table)
        g.reset_from_table()
        sims.append(g)
    a, b = sims
    a.rollout(130)
    b.rollout(130)          # mid-episode, first auto-resets behind us
    t = 130
    for n in chunks:
        for g in sims:
            g.step() if n == 1 else g.rollout(n)
        t += n
        assert nat.lib().cagpu_last_kernel().decode().startswith("ca_kernel<")      # (b ran last)
        _assert_same_bits(_state_bits(a), _state_bits(b), "after %d steps" % t)
    assert a.episode_stats()[0].item() > E // 4
    # a host write to the state without invalidate_plan() would leave a stale plan: the documented remedy works
    for g in sims:
        g.state["pos_x"].add_(0.01)
        g.invalidate_plan()
        g.step()
    _assert_same_bits(_state_bits(a), _state_bits(b), "after a host write")


def test_pipelined_kernel_mixed_policies_and_plan_entry_point():
    """non-RVO agents (external learner, non-cooperative, static) inside pipelined tiles; cagpu_plan on a reset state
    gives the plan the first step would compute"""
    nat, core, orc = _mods()
    N, E = 10, 64
    table = gu.fixtures(N)
    rng = np.random.default_rng(3)
    pol = rng.choice([nat.POL_RVO, nat.POL_RVO, nat.POL_NONCOOP, nat.POL_STATIC, nat.POL_LEARNING], size=(E, N))
    ext = rng.uniform(0, 1, (E, N, 2))
    sims = []
    for pl in (True, False):
        g = core.BatchedSim(core.make_params(E, N, game_over_mode=nat.OVER_ALL_DONE), record_actions=True, pipeline=pl)
        g.set_plugins(pol)
        g.set_fixture_table(table)
        g.reset_from_table()
        sims.append(g)
    a, b = sims
    assert a.try_plan()
    assert (a.state["flags"].cpu().numpy() >> 17 & 1).all()
    plan0 = a.state["next_action"].cpu().numpy().copy()
    for t in range(120):
        for g in sims:
            g.step(ext)
        if t == 0:
            rvo = pol == nat.POL_RVO
            assert np.array_equal(a.actions.cpu().numpy()[rvo], plan0[..., :2][rvo])      # the plan WAS the first action
        _assert_same_bits(_state_bits(a), _state_bits(b), "mixed policies, step %d" % t)


# ---------------------------------------------------------------- round 3: the reference's own full test suite on the GPU
@pytest.mark.parametrize("name", ["n10", "n4", "ragged4"])
def test_reference_suite_outcomes(name):
    """tests/golden/suite_*.npz: the unmodified reference's run_episode rows for 500 fixture cases (oracle/gen_suite_golden.py).
    Every case whose agents do not swap places exactly must end as in the reference -- outcome, step count, time to
    goal, final flags -- and the aggregate rates of all 500 must agree within a few episodes (the swap cases are
    symmetric head-on encounters decided by the last bit of atan2, where ROCm's libm and glibc differ: DESIGN.md section 5)."""
    nat, core, orc = _mods()
    ref = gu.load_suite(name)
    cases = gu.suite_cases(name)
    E, N = cases.shape[:2]
    g = core.BatchedSim(core.make_params(E, N, ragged=int(name == "ragged4")))
    g.set_plugins(nat.POL_RVO)
    g.reset(cases)
    if name == "ragged4":
        assert np.array_equal((g.state["flags"].cpu().numpy() >> 16 & 1).sum(1), 4 - ref["num_agents"])

    def step():
        g.step()
        return g.game_over.cpu().numpy(), {k: g.state[k].cpu().numpy() for k in ("t", "slt", "ep_reward", "pos_x", "pos_y")}

    got = gu.run_suite(g, cases, lambda: g.state["flags"].cpu().numpy().astype(np.uint32), step)
    swap = _swap_cases(cases)
    same = (got["outcome"] == ref["outcome"]) & (got["steps"] == ref["steps"]) & (got["flags"] == ref["flags"]).all(1) & \
           (np.abs(got["time_to_goal"] - ref["time_to_goal"]).max(1) < 1e-6) & \
           (np.abs(got["pos"] - ref["pos"]).max((1, 2)) < 1e-3)
    assert same[~swap].all(), "cases %s differ from the reference without an exact swap in them" % np.nonzero(~same & ~swap)[0][:20]
    assert same.mean() > 0.9
    for oc in range(3):
        assert abs(int((got["outcome"] == oc).sum()) - int((ref["outcome"] == oc).sum())) <= 12, (oc, got["outcome"], ref["outcome"])
    assert abs(got["steps"].mean() - ref["steps"].mean()) < 0.05 * ref["steps"].mean()


# ---------------------------------------------------------------- round 3: ragged batches (per-env agent counts)
@pytest.mark.parametrize("N,E,K", [(4, 600, 3), (10, 512, 9), (10, 512, 4), (20, 60, 19), (6, 77, 8)])
def test_ragged_batches_vs_oracle_reinjected(N, E, K):
    """envs with fewer agents than num_agents (CA_ABSENT slots: a case row with radius 0): neighbours, collisions,
    sensing, num_other_agents, zero padding, game over and episode statistics follow the env's OWN agent list like the
    reference's loops over len(self.agents); re-injected from the oracle every step, with auto-reset from a ragged table."""
    nat, core, orc = _mods()
    rng = np.random.default_rng(N * 100 + E)
    table = gu.fixtures(N).copy()
    n_e = rng.integers(2, N + 1, size=table.shape[0])
    for c in range(table.shape[0]):
        table[c, n_e[c]:] = 0.0
    o, g = _pair(E, N, K=K, ragged=1)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    cases = table[np.arange(E) % table.shape[0]]
    o.reset(cases)
    g.reset(cases)
    _compare_reset(o, g)
    o.rollout(table, 70)
    for t in range(90):
        _upload(o, g)
        o.rollout(table, 1)
        g.step()
        _compare(o, g, what="ragged N=%d step %d" % (N, t))
        np.testing.assert_allclose(g.state["env_stats"].cpu().numpy(), o.s["env_stats"], rtol=0, atol=1e-6)
    assert o.s["env_stats"][:, 0].sum() > E // 4
    absent = (o.view("flags") >> 16 & 1).astype(bool)
    assert absent.any() and not g.obs.cpu().numpy()[absent].any() and g.done.cpu().numpy()[absent].all()


# ---------------------------------------------------------------- the lean divide / square root sequences (csrc/cagpu.hip divq .. sqrtd)
def _rand_operands(rng, n, emin, emax):
    return np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(emin, emax + 1, n)) * rng.choice([-1.0, 1.0], n)


def test_lean_divide_and_sqrt_operand_range():
    """The ORCA phases and the distance / ego-frame code of the step kernels use the compiler's correctly rounded divide and
    square root sequences WITHOUT their range handling (no v_div_scale / v_div_fixup, no 2^32 rescaling).  Inside the
    documented operand range they are the IEEE results bit for bit -- that is what lets the step kernel's ORCA be compared
    bitwise with the oracle -- and this test pins the range: float exponents in [-60, 60] (ORCA's operands are 0 or
    1e-16 .. 1e8 in magnitude: velocities, positions, their differences and products), float64 exponents in [-300, 300].
    Beyond it (denormal or overflowing operands / quotients) the lean forms DO diverge: the documented limit of the step
    kernels (coordinates below 1e8 m), shown here so that it cannot go unnoticed."""
    nat, core, orc = _mods()
    rng = np.random.default_rng(42)
    n = 1 << 20
    bits = lambda x: np.ascontiguousarray(x).view(np.uint64)
    f32 = lambda x: x.astype(np.float32).astype(np.float64)
    # ---- inside the range: identical bits
    a, b = f32(_rand_operands(rng, n, -60, 60)), f32(_rand_operands(rng, n, -60, 60))
    lean, ieee = nat.debug_libm(2, a, b)
    assert np.array_equal(bits(lean), bits(ieee)) and np.array_equal(ieee, f32(a.astype(np.float32) / b.astype(np.float32)))
    lean, ieee = nat.debug_libm(3, np.abs(f32(_rand_operands(rng, n, -80, 80))))
    assert np.array_equal(bits(lean), bits(ieee))
    a, b = _rand_operands(rng, n, -300, 300), _rand_operands(rng, n, -300, 300)
    lean, ieee = nat.debug_libm(4, a, b)
    assert np.array_equal(bits(lean), bits(ieee)) and np.array_equal(ieee, a / b)
    x = np.abs(_rand_operands(rng, n, -600, 600))
    lean, ieee = nat.debug_libm(5, x)
    assert np.array_equal(bits(lean), bits(ieee)) and np.array_equal(ieee, np.sqrt(x))
    # the special operands ORCA does produce: exact zeros
    z = np.zeros(4)
    assert np.array_equal(nat.debug_libm(2, z, np.array([1.0, -2.0, 3.5, 1e8]))[0], z)
    assert np.array_equal(nat.debug_libm(3, z)[0], z) and np.array_equal(nat.debug_libm(5, z)[0], z)
    # ---- beyond the range: the correctly rounded forms stay IEEE, the lean ones do not
    report = {}
    with np.errstate(all="ignore"):
        for name, op, a, b in (
                ("float quotient overflows / denormal divisor", 2, f32(_rand_operands(rng, 4096, 60, 120)),
                 f32(_rand_operands(rng, 4096, -140, -110))),
                ("float denormal quotient", 2, f32(_rand_operands(rng, 4096, -100, -70)), f32(_rand_operands(rng, 4096, 40, 60))),
                ("float sqrt of a denormal", 3, np.abs(f32(_rand_operands(rng, 4096, -148, -128))), None),
                ("float64 quotient overflows", 4, _rand_operands(rng, 4096, 600, 1000), _rand_operands(rng, 4096, -1000, -600)),
                ("float64 denormal divisor", 4, _rand_operands(rng, 4096, -10, 10), _rand_operands(rng, 4096, -1070, -1030))):
            lean, ieee = nat.debug_libm(op, a, b)
            want = {2: lambda: f32(a.astype(np.float32) / b.astype(np.float32)), 3: lambda: f32(np.sqrt(a.astype(np.float32))),
                    4: lambda: a / b}[op]()
            same_ieee = (bits(ieee) == bits(want)) | (np.isnan(ieee) & np.isnan(want))
            report[name] = {"ieee_form_correct": float(same_ieee.mean()),
                            "lean_equals_ieee": float(((bits(lean) == bits(ieee)) | (np.isnan(lean) & np.isnan(ieee))).mean())}
    out = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out):
        import json
        json.dump(report, open(os.path.join(out, "lean_range.json"), "w"), indent=1)
    # measured (MI355X, ROCm 7.2): the IEEE forms are correct in every class; the lean forms agree on NO operand of the
    # overflowing / denormal-divisor / denormal-radicand classes and still agree on denormal QUOTIENTS of normal operands
    assert all(v["ieee_form_correct"] == 1.0 for v in report.values()), report
    for name in ("float quotient overflows / denormal divisor", "float sqrt of a denormal", "float64 quotient overflows",
                 "float64 denormal divisor"):
        assert report[name]["lean_equals_ieee"] < 0.01, report   # the limit is real: beyond the range they diverge


# ---------------------------------------------------------------- RVOPolicy's stochastic branches as per-step inputs
@pytest.mark.parametrize("N,E", [(10, 300), (4, 100), (20, 60)])
def test_rvo_stochastic_inputs_vs_oracle(N, E):
    """CaState.rvo_collab / rvo_heading_noise (RVOPolicy.py:77-90, :118-119: the ego's collaboration coefficient of each
    query, the noise added to the clipped delta heading): the same draws handed to the kernel and to the oracle give the
    same ORCA velocities bit for bit and the same step; with them the policy is queried at the start of the step (the
    unpipelined kernel), although the sim carries next_action."""
    nat, core, orc = _mods()
    rng = np.random.default_rng(N * E)
    table = gu.fixtures(N)
    o, g = _pair(E, N)
    o.s["policy"][:] = orc.POL_RVO
    g.set_plugins(nat.POL_RVO)
    o.reset(table[np.arange(E) % table.shape[0]])
    for _ in range(40):
        o.step()
    keep = []
    for t in range(12):
        collab = rng.choice(np.array([0.0, -0.5, 0.5, -1.0], np.float32), (E, N))
        noise = rng.normal(0.0, 0.5, (E, N)) * (rng.random((E, N)) < 0.5)
        o.set_rvo_stochastic(collab, noise)
        tc_, tn_ = torch.from_numpy(collab).to(g.device), torch.from_numpy(noise).to(g.device)
        keep = [tc_, tn_]
        g._cs.rvo_collab, g._cs.rvo_heading_noise = tc_.data_ptr(), tn_.data_ptr()
        _upload(o, g)
        o.step()
        g.step()
        assert nat.lib().cagpu_last_kernel().decode().startswith("ca_kernel<")
        got, want = g.orca_vel.cpu().numpy().reshape(-1, 2), o.orca_vel.reshape(-1, 2)
        # (equal as NUMBERS: with a collaboration coefficient of exactly 0 -- the non-cooperative phase -- the products
        # 0 * u carry u's sign, and the kernel's branch-free half-plane and the oracle's if / else form can hand an agent
        # whose solution is the zero velocity zeros of different sign: seen for ~1 agent in 10 000, without consequence --
        # the position advance pos + v * dt and every later use are the same)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "step %d: %d ORCA velocities differ; first: agent %s collab %s got %s want %s flags %s" % (
            t, bad.size, bad[:6], collab.reshape(-1)[bad[:6]], got[bad[:6]], want[bad[:6]],
            [hex(int(f)) for f in o.s["flags"][bad[:6]]])
        _compare(o, g, what="stochastic N=%d step %d" % (N, t))
    # the draws matter: an agent's action with noise differs from the one without
    assert np.abs(g.actions.cpu().numpy()[..., 1]).max() > np.pi / 6 + 0.05
    g._cs.rvo_collab, g._cs.rvo_heading_noise = None, None
    o.set_rvo_stochastic()
    _upload(o, g)
    o.step()
    g.step()
    _compare(o, g, what="deterministic again")
    assert np.abs(g.actions.cpu().numpy()[..., 1]).max() <= np.pi / 6 + 1e-6


def test_rvo_stochastic_batch_statistics():
    """core.BatchedSim.set_rvo_stochastic: the batched device draws follow the reference's rules -- an anti-collaborative
    agent (RVO_COLLAB_COEFF = c < 0) redraws `use_non_coop_policy` whenever its clock is within DT of a multiple of
    RVO_ANTI_COLLAB_T (True with probability 1 - |c|, then collaboration coefficient 0, else c); heading noise is
    N(0, 0.5) on the agents that have it and exactly 0 on the others"""
    nat, core, orc = _mods()
    E, N, c, T = 3000, 10, -0.7, 1.0
    g = core.BatchedSim(core.make_params(E, N), record_actions=True)
    g.set_plugins(nat.POL_RVO)
    table = gu.fixtures(N)
    g.reset(table[np.arange(E) % 500])
    mask = np.zeros((E, N), bool)
    mask[:, ::2] = True
    g.set_rvo_stochastic(heading_noise=mask, collab_coeff=c, anti_collab_t=T, seed=11)
    assert bool(g._rvo["non_coop"].all())                    # RVOPolicy.py:33: starts non-cooperative
    prev, switched = None, {}
    noises = []
    for t in range(25):
        g.step()
        assert nat.lib().cagpu_last_kernel().decode().startswith("ca_kernel<")
        nc = g._rvo["non_coop"].cpu().numpy()
        cf = g._rvo["collab"].cpu().numpy()
        assert np.array_equal(cf, np.where(nc, np.float32(0.0), np.float32(c)))
        if prev is not None:
            switched[t] = float((nc != prev).mean())
        prev = nc
        noises.append(g._rvo["noise"].cpu().numpy())
        if t in (0, 10, 20):                                  # t = 0, 1, 2 s: a redraw for everybody
            assert abs(nc.mean() - (1 - abs(c))) < 0.02, (t, nc.mean())
    # (step t queries at clock t * DT: everybody redraws at t = 10 and 20 -- a switch with probability 2 * 0.3 * 0.7 --; in
    # between only an agent whose clock STOPPED on a multiple of T -- done since that step -- keeps redrawing)
    assert all(abs(switched[t] - 0.42) < 0.03 for t in (10, 20)), switched
    assert all(v < 0.01 for t, v in switched.items() if t not in (10, 20)), switched
    z = np.stack(noises)
    assert np.all(z[:, ~mask] == 0.0)
    assert abs(z[:, mask].mean()) < 0.005 and abs(z[:, mask].std() - 0.5) < 0.005
    # the noise reaches the actions: unclipped delta headings beyond pi / 6 appear only on the noisy agents
    a1 = np.abs(g.actions.cpu().numpy()[..., 1])
    assert a1[:, 1::2].max() <= np.pi / 6 + 1e-6 and a1[:, ::2].max() > np.pi / 6 + 0.1
    assert np.isfinite(g.state["pos_x"].cpu().numpy()).all()
    g.set_rvo_stochastic()
    g.step()
    assert nat.lib().cagpu_last_kernel().decode().startswith("ca_pipe_kernel<10, 4")


def test_ext_state_applied_at_the_move_vs_oracle():
    """CaState.ext_state: agents with ExternalDynamics take an externally integrated state AT THE MOVE of the step -- after
    every policy has seen the pre-step state, before at-goal / clocks / collisions / sensing -- NaN rows and agents with
    built-in dynamics ignore it; kernel and oracle agree (re-injected)"""
    nat, core, orc = _mods()
    N, E = 6, 120
    rng = np.random.default_rng(9)
    table = gu.fixtures(N)
    o, g = _pair(E, N)
    pol = np.where(rng.random((E, N)) < 0.6, orc.POL_RVO, orc.POL_NONCOOP).astype(np.int32)
    dyn = np.where(rng.random((E, N)) < 0.4, orc.DYN_EXTERNAL, orc.DYN_UNICYCLE).astype(np.int32)
    o.set_policies(pol, dyn)
    g.set_plugins(pol, dyn)
    o.reset(table[np.arange(E) % 500])
    for _ in range(15):
        o.step()
    for t in range(12):
        st = np.full((E, N, 5), np.nan)
        move = rng.random((E, N)) < 0.7
        cur = np.stack([o.view(n) for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading")], axis=-1)
        st[move] = cur[move] + rng.normal(0, 0.05, (int(move.sum()), 5))
        o.set_ext_state(st)
        _upload(o, g)
        o.step()
        g.step(ext_state=st)
        assert nat.lib().cagpu_last_kernel().decode().startswith("ca_kernel<")
        _compare(o, g, what="ext_state step %d" % t)
    took = (dyn == orc.DYN_EXTERNAL) & move & ((o.view("flags") & (orc.AT_GOAL | orc.OUT_OF_TIME | orc.IN_COLLISION)) == 0)
    assert took.sum() > 50
    np.testing.assert_allclose(g.state["pos_x"].cpu().numpy()[took], st[..., 0][took], rtol=0, atol=1e-12)
    g.step()            # without ext_state again: the pointer is cleared
    assert g._cs.ext_state is None or not g._cs.ext_state


# ---------------------------------------------------------------- more than 64 agents per env (csrc/cagpu_big.inc)
@pytest.mark.parametrize("N,E,K,sort,ragged", [(100, 5, 99, 0, 0), (70, 9, 19, 1, 1), (65, 4, 10, 2, 0), (128, 3, 30, 0, 0),
                                               (300, 2, 19, 0, 0), (600, 1, 12, 1, 1)])
def test_big_envs_vs_oracle(N, E, K, sort, ragged):
    """Envs beyond one workgroup tile -- the reference's make_testcase_huge / get_testcase_huge scenes (test_cases.py:914-1018;
    100 agents in its shipped case) -- run the one-thread-per-agent kernel over CaOut.workspace: reset, re-injected steps with
    every built-in policy / dynamics, all three sort modes, ragged slots, auto-reset from a table, against the oracle; the
    ORCA velocities bit for bit (the serial programme of cagpu_orca)."""
    nat, core, orc = _mods()
    from gym_collision_avoidance_amd.envs import test_cases as tc
    rng = np.random.default_rng(N + E)
    np.random.seed(N)
    # (a quarter of the square covered by the 2 m clearance discs: the rejection sampler of make_testcase_huge stays quick)
    table = tc.make_testcase_huge(E + 3, N, side_length=2.0 * np.sqrt(N) + 3.0, speed_bnds=[0.5, 1.5], radius_bnds=[0.2, 0.5])
    if ragged:
        for c in range(table.shape[0]):
            table[c, N - int(rng.integers(0, N // 3)):] = 0.0       # (padding rows: radius 0 = empty slots)
    # (game over when agent 0 is done and short clocks: time-outs, goals and auto-resets all fall into the compared window)
    mtr = 0.25 if N <= 128 else 0.04      # (the larger scenes have longer trips: agent 0 still runs out of time inside the window)
    o, g = _pair(E, N, K, sort_mode=sort, ragged=ragged, max_time_ratio=mtr, game_over_mode=1)
    assert g._workspace is not None and g._workspace.numel() == int(nat.lib().cagpu_workspace_bytes(g.p))
    pol = rng.choice([orc.POL_RVO, orc.POL_RVO, orc.POL_RVO, orc.POL_NONCOOP, orc.POL_STATIC, orc.POL_EXTERNAL,
                      orc.POL_LEARNING], (E, N)).astype(np.int32)
    dyn = rng.choice([orc.DYN_UNICYCLE, orc.DYN_UNICYCLE, orc.DYN_MAX_TURN_RATE], (E, N)).astype(np.int32)
    o.set_policies(pol, dyn)
    g.set_plugins(pol, dyn)
    g.set_fixture_table(table)
    o.reset(table[:E])
    g.reset(table[:E])
    assert nat.lib().cagpu_last_kernel().decode().startswith("ca_big_kernel")
    _compare_reset(o, g)
    ended = 0
    for t in range(45):
        ext = rng.uniform(0.0, 1.0, (E, N, 2))
        _upload(o, g)
        o.rollout_ex(table, 1, ext_actions=ext)
        g.step(ext)
        assert nat.lib().cagpu_last_kernel().decode().startswith("ca_big_kernel")
        live = (o.s["policy"] == orc.POL_RVO)
        got, want = g.orca_vel.cpu().numpy().reshape(-1, 2), o.orca_vel.reshape(-1, 2)
        assert np.array_equal(got, want), "step %d: %d ORCA velocities differ" % (t, (got != want).any(1).sum())
        _compare(o, g, what="N=%d step %d" % (N, t))
        ended += int(o.game_over.sum())
    assert ended > 0 and live.any()
    # observe() and a fused rollout (n launches of the same kernel) on the same path
    before = g.obs.clone()
    g.obs.zero_()
    assert torch.equal(g.observe(), before)
    h = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=sort, ragged=ragged, max_time_ratio=mtr, game_over_mode=1))
    h.set_plugins(np.where(pol >= orc.POL_EXTERNAL, orc.POL_NONCOOP, pol), dyn)
    h.set_fixture_table(table)
    g.set_plugins(np.where(pol >= orc.POL_EXTERNAL, orc.POL_NONCOOP, pol), dyn)
    for n in F64 + ("flags", "step_num", "episode_step", "reset_count", "last_action"):
        h.state[n].copy_(g.state[n])
    h.set_plugins(np.where(pol >= orc.POL_EXTERNAL, orc.POL_NONCOOP, pol), dyn)
    g.rollout(4)
    for _ in range(4):
        h.step()
    for n in ("pos_x", "pos_y", "heading", "flags"):
        assert torch.equal(g.state[n], h.state[n]), n


def test_big_envs_through_the_env_api():
    """the reference's get_testcase_huge shape through the gym-level API: 100 RVO agents in one env"""
    from tests import envtools
    Config, tc, Env = envtools.fresh("Huge100")
    np.random.seed(5)
    case = tc.make_testcase_huge(1, 100, side_length=25)[0]      # (the reference's own default side length)
    env = Env()
    env.set_agents(tc.cadrl_test_case_to_agents(case, policies="RVO"))
    obs, _ = env.reset()
    assert obs[0]["other_agents_states"].shape == (Config.MAX_NUM_OTHER_AGENTS_OBSERVED, 7)
    d0 = np.array([a.dist_to_goal for a in env.agents])
    for _ in range(40):
        obs, rew, over, _, info = env.step({})
    d1 = np.array([a.dist_to_goal for a in env.agents])
    assert rew.shape == (100,) and (d1 < d0 - 1.0).mean() > 0.8 and len(info["which_agents_done"]) == 100


@pytest.mark.parametrize("N,E,K,sort,ragged", [(20, 300, 19, 1, 0), (20, 64, 19, 0, 0), (6, 400, 9, 1, 0), (10, 100, 5, 1, 1),
                                               (32, 20, 24, 1, 0)])
def test_ga3c_fused_sensing_equals_the_stored_observation(N, E, K, sort, ragged):
    """BASELINE configs[2] as worded -- "ego-centric obs + network inference fused in-kernel": cagpu_ga3c with obs = NULL
    computes, for exactly the agents it evaluates, OtherAgentsStatesSensor.sense + the observation assembly from the state
    arrays inside the network kernel.  The float32 values that reach the LSTM must be the ones the stored observation row
    holds: logits and choices of the fused call equal those of the call that reads the rows, BIT FOR BIT, along episodes
    (clip < N - 1, K < 19, K > 19, closest_first / closest_last, ragged slots, agents that are done)."""
    nat, core, orc = _mods()
    from gym_collision_avoidance_amd.envs import test_cases as tc
    rng = np.random.default_rng(N * E)
    if N in (6, 10, 20):
        table = tc.fixture_table(N).astype(np.float32).astype(np.float64)
    else:
        np.random.seed(N)
        table = tc.make_testcase_huge(40, N, side_length=14.0, speed_bnds=[0.5, 1.5], radius_bnds=[0.2, 0.5])
    if ragged:
        table = table.copy()
        for c in range(table.shape[0]):
            table[c, N - int(rng.integers(0, 4)):] = 0.0
    g = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=sort, ragged=ragged, obs_clip=min(K, 7) if N == 10 else None))
    pol = np.full((E, N), nat.POL_GA3C_CADRL)
    pol[:, 1::4] = nat.POL_RVO
    g.set_plugins(pol)
    g.load_ga3c(keep_logits=True)
    g.set_fixture_table(table)
    g.reset_from_table()
    compared = 0
    for t in range(30):
        ext_a = torch.full((E, N, 2), -7.0, dtype=torch.float64, device=g.device)
        ext_b = ext_a.clone()
        g.ga3c_logits.fill_(-555.0)
        g.ga3c(ext_a, fused=False)
        la = g.ga3c_logits.clone()
        g.ga3c_logits.fill_(-555.0)
        g.ga3c(ext_b, fused=True)
        assert torch.equal(la, g.ga3c_logits), "step %d: %d logits differ (max %g)" % (
            t, int((la != g.ga3c_logits).sum()), float((la - g.ga3c_logits).abs().max()))
        assert torch.equal(ext_a, ext_b)
        compared += int((ext_a[..., 0] >= 0).sum())
        g.step()
    assert compared > 200
