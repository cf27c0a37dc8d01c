"""GPU parity AT THE BENCH GEOMETRIES: every workload bench.py can time is built through bench.build_workload (so the
kernel instantiation, tile size and grid are the timed ones), runs on the GPU at FULL size, and blocks of envs sampled
from the batch are stepped by the CPU oracle from the GPU's own state -- envs never interact, so a block of a 32 768-env
batch is a complete simulation.  The GPU free-runs (its pipelined plan stays valid: the steady-state code path of the
benchmark); the oracle is re-seeded from the GPU's bits every step.  Bars as everywhere: masks exact, floats 1e-5.

Also here: the demonstration that the free-running divergence of the exact-swap fixture cases is caused by the libm bits
alone (test_swap_cases_agree_on_the_gpus_libm_bits)."""
import math
import os
import struct
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from tests import golden_util as gu  # noqa: E402
from tests.test_gpu_parity import F64, MASK, TOL, _mods, _swap_cases  # noqa: E402

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- block helpers
class Block(object):
    """envs [b0, b0 + n) of a GPU batch mirrored in a small oracle"""

    def __init__(self, g, b0, n, table, policy, env_id_offset=0, static_map=None):
        nat, core, orc = _mods()
        p = g.p
        self.g, self.b0, self.n, self.table = g, int(b0), int(n), table
        self.sl = slice(self.b0, self.b0 + self.n)
        self.off = int(env_id_offset) + self.b0
        self.o = orc.Oracle(orc.default_params(self.n, g.N, max_obs=g.K, sort_mode=p.sort_mode))
        self.o.set_policies(policy)
        if static_map is not None:
            self.o.set_map(static_map)
        self.first = True

    def download(self, with_obs=False):
        """oracle block := GPU block (both continue from identical bits)"""
        nat, core, orc = _mods()
        g, o, sl = self.g, self.o, self.sl
        for n in F64:
            o.s[n][:] = g.state[n][sl].cpu().numpy().reshape(-1)
        o.s["last_action"][:] = g.state["last_action"][sl].cpu().numpy().reshape(o.s["last_action"].shape)
        fl = g.state["flags"][sl].cpu().numpy().reshape(-1).astype(np.uint32)
        o.s["flags"][:] = (fl & 0xFF) | (fl & orc.ABSENT)
        o.s["step_num"][:] = g.state["step_num"][sl].cpu().numpy().reshape(-1)
        o.s["episode_step"][:] = g.state["episode_step"][sl].cpu().numpy()
        o.s["reset_count"][:] = g.state["reset_count"][sl].cpu().numpy()
        o.s["env_stats"][:] = g.state["env_stats"][sl].cpu().numpy()
        if with_obs:
            o.obs[:] = g.obs[sl].cpu().numpy().astype(np.float64)
        if g.scan is not None:
            o.scan_hist[:] = g.scan_hist[sl].cpu().numpy()

    def step(self):
        self.o.rollout_ex(self.table, 1, env_id_offset=self.off, case_stride=self.g._ar.case_stride)

    def compare(self, what, rows=None):
        """masks exact, floats within TOL; rows: bool [n] -- compare only these envs (GA3C: where the choices agree)"""
        g, o, sl = self.g, self.o, self.sl
        rows = np.ones(self.n, bool) if rows is None else rows
        ra = np.repeat(rows, g.N)
        gf = g.state["flags"][sl].cpu().numpy().reshape(-1).astype(np.uint32)
        assert np.array_equal((gf & MASK)[ra], (o.s["flags"] & MASK)[ra]), "flags " + what
        assert np.array_equal(g.done[sl].cpu().numpy()[rows], o.done[rows]), "done " + what
        assert np.array_equal(g.game_over[sl].cpu().numpy()[rows], o.game_over[rows]), "game_over " + what
        for n in F64:
            a, b = g.state[n][sl].cpu().numpy().reshape(-1)[ra], o.s[n][ra]
            if n == "heading":          # an angle: compared modulo 2 pi (see tests/test_gpu_parity.py::_compare)
                d = np.abs((a - b + np.pi) % (2 * np.pi) - np.pi)
                assert d.size == 0 or d.max() <= TOL, "heading %s: %g" % (what, d.max())
            elif n == "turning_dir":    # branches on the sign of a heading that may sit on the wrap boundary or at 0
                bad = np.abs(a - b) > TOL
                ho, hg = o.s["heading"][ra], g.state["heading"][sl].cpu().numpy().reshape(-1)[ra]
                edge = (np.abs(ho) < 1e-7) | (np.abs(np.abs(ho) - np.pi) < 1e-7) | (np.abs(hg) < 1e-7) | (np.abs(np.abs(hg) - np.pi) < 1e-7)
                assert not (bad & ~edge).any(), "turning_dir %s: differs away from a sign change of the heading" % what
            else:
                np.testing.assert_allclose(a, b, rtol=0, atol=TOL, err_msg=n + " " + what)
        gobs = g.obs[sl].cpu().numpy().astype(np.float64)[rows]
        assert np.array_equal(gobs[..., 1], o.obs[rows][..., 1]), "num_other_agents " + what
        np.testing.assert_allclose(gobs, o.obs[rows], rtol=0, atol=TOL, err_msg="obs " + what)
        np.testing.assert_allclose(g.rewards[sl].cpu().numpy()[rows], o.rewards[rows], rtol=0, atol=TOL, err_msg="rewards " + what)
        for n, a in (("step_num", o.s["step_num"].reshape(self.n, g.N)), ("episode_step", o.s["episode_step"]),
                     ("reset_count", o.s["reset_count"])):
            assert np.array_equal(g.state[n][sl].cpu().numpy()[rows], a[rows]), n + " " + what
        np.testing.assert_allclose(g.state["env_stats"][sl].cpu().numpy()[rows], o.s["env_stats"][rows], rtol=0, atol=1e-6,
                                   err_msg="env_stats " + what)


def _blocks(E, n, count, seed):
    """`count` blocks of n envs: the first tile, the last envs of the batch, the rest at random"""
    rng = np.random.default_rng(seed)
    starts = {0, max(0, E - n)}
    while len(starts) < min(count, max(1, E // n)):
        starts.add(int(rng.integers(0, E - n + 1)))
    return sorted(starts)


def _bench():
    import bench
    return bench


def _run_rvo(E, steps, warm, expect_kernel, n_block=16, count=6):
    nat, core, orc = _mods()
    sim, table, N, K = _bench().build_workload("rvo10", E, torch.device("cuda", 0))
    sim.rollout(warm)           # mid-episode: time-outs, goals, collisions and auto-resets all fall into the window
    blocks = [Block(sim, b0, n_block, table, orc.POL_RVO) for b0 in _blocks(E, n_block, count, E)]
    ended = 0
    for t in range(steps):
        for b in blocks:
            b.download()
        sim.step()
        kern = nat.lib().cagpu_last_kernel().decode()
        assert kern.startswith(expect_kernel), kern
        for b in blocks:
            b.step()
            b.compare("E=%d block %d step %d" % (E, b.b0, t))
            ended += int(b.o.game_over.sum())
    assert ended > 0, "no episode ended inside the compared window"
    return sim


# ---------------------------------------------------------------- configs 2, 4 and the metric size
def test_config2_1024x10_vs_oracle():
    """BASELINE configs[1]: 1024 envs x 10 RVO agents on one GPU, the instantiation and grid the launcher picks there"""
    nat, core, orc = _mods()
    _run_rvo(1024, steps=30, warm=150, expect_kernel="ca_pipe_kernel<10, ")
    # (the tile size of this batch size is whatever launch_pipe chose: it is part of the name the bench line reports)
    assert "grid=" in nat.lib().cagpu_last_kernel().decode()


def test_metric_4096x10_free_running_vs_oracle_blocks():
    """the metric's own batch with VALID plans (the steady state bench.py times: test_metric_geometry_step_vs_oracle
    re-injects the oracle's state, which invalidates the plan every step)"""
    _run_rvo(4096, steps=30, warm=150, expect_kernel="ca_pipe_kernel<10, 4, false> grid=1024")


def test_config4_32768x10_vs_oracle():
    """BASELINE configs[3] as ONE batch on one GPU (the driver shards it 8 x 4096; this is the harder geometry: eight rounds
    of resident tiles)"""
    _run_rvo(32768, steps=12, warm=150, expect_kernel="ca_pipe_kernel<10, 4, false> grid=8192", count=8)


def test_sharded_block_equals_single_batch():
    """a shard of config 4 (rank 5 of 8 x 4096) replays exactly the case streams of envs 20480 .. 24575 of the single batch"""
    nat, core, orc = _mods()
    dev = torch.device("cuda", 0)
    whole, table, N, K = _bench().build_workload("rvo10", 32768, dev)
    shard, _, _, _ = _bench().build_workload("rvo10", 4096, dev, rank=5, world=8)
    for _ in range(4):
        whole.rollout(60)
        shard.rollout(60)
        sl = slice(5 * 4096, 6 * 4096)
        for n in F64 + ("flags", "reset_count", "episode_step"):
            assert torch.equal(whole.state[n][sl], shard.state[n]), n
        assert torch.equal(whole.obs[sl], shard.obs)
    assert int(shard.state["reset_count"].sum()) > 2048     # (240 steps: most envs are in their second episode)


# ---------------------------------------------------------------- config 3
def test_config3_4096x20_ga3c_vs_oracle():
    """BASELINE configs[2]: 4096 x 20 GA3C-CADRL agents, K = 19, closest_last.  Per sampled env: the network's choice per
    agent against the numpy restatement on the SAME observation row (the GPU's), and masks / state / observation of the
    step wherever all of an env's choices agree (a choice may differ where two logits tie within fp32 round-off)."""
    nat, core, orc = _mods()
    E = 4096
    sim, table, N, K = _bench().build_workload("ga3c20", E, torch.device("cuda", 0))
    sim.load_ga3c(keep_logits=True)
    for _ in range(60):
        sim.step()
    blocks = [Block(sim, b0, 8, table, orc.POL_GA3C_CADRL) for b0 in _blocks(E, 8, 5, 3)]
    asked = agree = envs_cmp = ended = 0
    for t in range(25):
        for b in blocks:
            b.download(with_obs=True)
        sim.step()
        kern = nat.lib().cagpu_last_kernel().decode()
        assert kern.startswith("ca_kernel<256, ") and ", 20, " in kern, kern
        gi = sim._ga3c_ext.cpu().numpy()[..., 0]
        lg = np.sort(sim.ga3c_logits.cpu().numpy(), axis=-1)
        for b in blocks:
            b.step()
            q = b.o.ga3c_index.reshape(b.n, N)
            was_asked = q >= 0
            same = (gi[b.sl] == q) | ~was_asked
            margin = (lg[b.sl][..., -1] - lg[b.sl][..., -2])
            assert np.all(margin[~same] < 1e-3), "step %d block %d: a clear-cut choice differs" % (t, b.b0)
            asked += int(was_asked.sum())
            agree += int((same & was_asked).sum())
            ok = same.all(axis=1)
            envs_cmp += int(ok.sum())
            b.compare("ga3c block %d step %d" % (b.b0, t), rows=ok)
            ended += int(b.o.game_over.sum())
    assert asked > 2000 and agree >= 0.995 * asked, (agree, asked)
    assert envs_cmp >= 0.9 * 25 * sum(b.n for b in blocks)


def test_config3_episode_outcomes_free_running_vs_oracle():
    """What the argmax choices that differ between the kernel's network and the numpy one (< 0.5 %, all within 1e-3 of a tie)
    do to EPISODES: 2048 twenty-agent GA3C-CADRL scenes (drawn by cagpu_generate_cases, sides 6 .. 8 m like the n20 fixture)
    run to their end free-running on the GPU and, from the same cases, free-running in the oracle (C++ step + numpy network);
    collision / all-at-goal / stuck rates must agree within 3 sigma of two binomial samples of this size, and so must the
    mean episode length."""
    nat, core, orc = _mods()
    dev = torch.device("cuda", 0)
    E, N, K, T = 2048, 20, 19, 700
    sim = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=nat.SORT_CLOSEST_LAST), device=dev)
    sim.set_plugins(nat.POL_GA3C_CADRL, nat.DYN_UNICYCLE)
    sim.load_ga3c()
    cases = sim.generate_cases(E, seed=2024, side_length=(6.0, 8.0))
    sim.reset(cases)
    # the oracle side (C++ step + numpy network, ~2 s per 100 steps of 256 envs) on 16 processes of 128 envs each, started
    # first so that they run beside the GPU's part
    import multiprocessing as mp
    from tests import oracle_workers
    host_cases = cases.cpu().numpy()
    keep = {k_: os.environ.get(k_) for k_ in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update({k_: "4" for k_ in keep})      # (inherited by the workers: 16 processes x 4 BLAS threads)
    try:
        pool = mp.get_context("spawn").Pool(16)
    finally:
        for k_, v_ in keep.items():
            os.environ.pop(k_, None) if v_ is None else os.environ.__setitem__(k_, v_)
    pending = pool.map_async(oracle_workers.ga3c_episodes, [(host_cases[i:i + 128], T) for i in range(0, E, 128)])
    o1 = orc.Oracle(orc.default_params(128, N, max_obs=K, sort_mode=orc.SORT_CLOSEST_LAST))
    o1.set_policies(orc.POL_GA3C_CADRL)
    o1.reset(host_cases[:128])
    np.testing.assert_allclose(sim.obs[:128].cpu().numpy(), o1.obs, rtol=0, atol=1e-5)
    g_len = np.zeros(E)
    g_over = torch.zeros(E, dtype=torch.bool, device=dev)
    for t in range(T):
        sim.step()
        g_len += (~g_over).cpu().numpy()
        g_over |= sim.game_over.view(torch.bool)
        if bool(g_over.all()):
            break
    parts = pending.get(timeout=900)
    pool.close()
    o_flags = np.concatenate([p_[0] for p_ in parts])
    o_over = np.concatenate([p_[1] for p_ in parts])
    o_len = np.concatenate([p_[2] for p_ in parts])

    def outcomes(flags, over):
        f = flags.reshape(E, N)
        coll = ((f & (orc.IN_COLLISION | orc.WAS_IN_COLLISION)) != 0).any(axis=1)
        goal = ((f & orc.AT_GOAL) != 0).all(axis=1)
        return np.array([(coll & over).mean(), (goal & ~coll & over).mean(), (~coll & ~goal & over).mean(), (~over).mean()])
    pg = outcomes(sim.state["flags"].cpu().numpy().astype(np.uint32), g_over.cpu().numpy())
    po = outcomes(o_flags, o_over)
    names = ("collision", "all at goal", "stuck / out of time", "unfinished after %d steps" % T)
    for nme, a, b in zip(names, pg, po):
        bound = 3.0 * np.sqrt((a * (1 - a) + b * (1 - b)) / E) + 1.0 / E
        assert abs(a - b) <= bound, "%s: GPU %.4f oracle %.4f (3 sigma = %.4f)" % (nme, a, b, bound)
    assert pg[0] + pg[1] > 0.5 and po[3] < 0.2, (pg, po)      # (the window holds most episodes to their end)
    se = np.sqrt((g_len.var() + o_len.var()) / E)
    assert abs(g_len.mean() - o_len.mean()) <= 3.0 * se + 0.5, (g_len.mean(), o_len.mean(), se)
    same = (g_len == o_len).mean()
    print("config-3 episodes: GPU %s oracle %s; mean length %.1f / %.1f; %.1f %% of the episodes end in the same step" % (
        np.round(pg, 4), np.round(po, 4), g_len.mean(), o_len.mean(), 100 * same))


# ---------------------------------------------------------------- config 5
def test_config5_4096x50_map_laserscan_vs_oracle():
    """BASELINE configs[4]: 4096 x 50 RVO agents + static map (wall collisions) + 512-beam LaserScanSensor"""
    nat, core, orc = _mods()
    bench = _bench()
    E = 4096
    sim, table, N, K = bench.build_workload("crowd50_laser", E, torch.device("cuda", 0))
    sim.laserscan()
    for _ in range(40):
        sim.step()
        sim.laserscan()
    blocks = [Block(sim, b0, 2, table, orc.POL_RVO, static_map=bench.crowd_map()) for b0 in _blocks(E, 2, 4, 5)]
    bad = total = 0
    for t in range(10):
        for b in blocks:
            b.download()
        sim.step()
        kern = nat.lib().cagpu_last_kernel().decode()
        assert kern.startswith("ca_kernel<512, ") or kern.startswith("ca_kernel<256, "), kern
        scan = sim.laserscan().cpu().numpy()
        hist = sim.scan_hist.cpu().numpy()
        for b in blocks:
            b.step()
            b.compare("crowd block %d step %d" % (b.b0, t))
            want = np.rint(b.o.laserscan() / 0.1).astype(np.int64)
            got = np.rint(scan[b.sl].astype(np.float64) / 0.1).astype(np.int64)
            assert np.array_equal(hist[b.sl] == 255, want == 60)
            bad += int((got != want).sum())
            total += want.size
    assert total > 1e6 and bad <= max(3, total // 200000), "%d of %d beams differ" % (bad, total)
    assert (sim.state["flags"].cpu().numpy() & nat.IN_COLLISION).any()


# ---------------------------------------------------------------- the swap cases, on the GPU's own libm bits
class _GpuLibm(object):
    """atan2 / (sin, cos) tables filled by cagpu_debug_libm: the oracle's libm hooks look their operands up here; an
    operand not seen yet is noted (and answered with the host's libm) so that the caller can evaluate it on the device and
    repeat the step"""

    def __init__(self):
        self.at, self.sc, self.miss_at, self.miss_sc = {}, {}, [], []

    @staticmethod
    def _key(*v):     # by BITS: 0.0 == -0.0 as dict keys, but atan2(+0, -1) = pi and atan2(-0, -1) = -pi
        return struct.pack("%dd" % len(v), *v)

    def atan2(self, y, x):
        v = self.at.get(self._key(y, x))
        if v is None:
            self.miss_at.append((y, x))
            return math.atan2(y, x)
        return v

    def sincos(self, a):
        v = self.sc.get(self._key(a))
        if v is None:
            self.miss_sc.append(a)
            return math.sin(a), math.cos(a)
        return v

    def resolve(self):
        """evaluate the noted operands on the device; -> number resolved"""
        nat, core, orc = _mods()
        n = len(self.miss_at) + len(self.miss_sc)
        if self.miss_at:
            m = np.array(self.miss_at, np.float64)
            r, _ = nat.debug_libm(0, m[:, 0], m[:, 1])
            self.at.update({self._key(*k): float(v) for k, v in zip(self.miss_at, r)})
        if self.miss_sc:
            m = np.array(self.miss_sc, np.float64)
            s, c = nat.debug_libm(1, m)
            self.sc.update({self._key(k): (float(a), float(b)) for k, a, b in zip(self.miss_sc, s, c)})
        self.miss_at, self.miss_sc = [], []
        return n


def test_swap_cases_agree_on_the_gpus_libm_bits():
    """VERDICT r03 weak-1.  The ten-agent fixture cases in which two agents exchange places exactly meet head-on in a
    perfectly symmetric configuration; ORCA amplifies lateral round-off ~12x per step there, so a free-running GPU episode
    and a free-running oracle episode may take mirror-image paths (test_free_running_vs_oracle_10_agents excuses them).
    Claim: the ONLY seed of that divergence is the libm -- ROCm's atan2 and the kernels' short-range sin / cos against
    glibc's.  Demonstration: the oracle, FREE-RUNNING on its own state for whole episodes, with nothing but those two
    operations answered by the device (cagpu_debug_libm), reproduces the GPU's episodes of ALL swap cases -- positions,
    velocities, headings to the bit, flags and outcomes exactly -- while the same oracle on glibc's bits does not."""
    nat, core, orc = _mods()
    N = 10
    table = gu.fixtures(N)
    cases = table[_swap_cases(table)]
    E = cases.shape[0]
    assert 50 < E < 100
    T = 260           # (the head-on encounters happen within the first ~100 steps; a stuck episode may run for thousands)
    g = core.BatchedSim(core.make_params(E, N))
    g.set_plugins(nat.POL_RVO)
    g.reset(cases)
    traj = []
    for t in range(T):
        g.step()
        traj.append({n: g.state[n].cpu().numpy().reshape(-1).copy() for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading")})
        traj[-1]["flags"] = g.state["flags"].cpu().numpy().reshape(-1).astype(np.uint32) & MASK
    assert float(g.game_over.float().mean().item()) > 0.8, "most swap episodes should be over after %d steps" % T

    def oracle_run(libm):
        o = orc.Oracle(orc.default_params(E, N))
        o.s["policy"][:] = orc.POL_RVO
        if libm is not None:
            orc.set_libm(libm.atan2, libm.sincos)
        try:
            o.reset(cases)
            if libm is not None:          # the reset headings (test_cases.py:554) on the device's atan2 as well
                while libm.resolve():
                    o.reset(cases)
            worst, first_bad = 0.0, None
            for t in range(T):
                if libm is not None:
                    keep = {k: v.copy() for k, v in o.s.items()}
                    for _ in range(8):    # policy atan2 -> heading sincos -> ego-frame atan2: three dependent batches
                        o.step()
                        if not libm.resolve():
                            break
                        for k, v in keep.items():
                            o.s[k][:] = v
                    else:
                        raise AssertionError("libm table did not converge")
                else:
                    o.step()
                for n in ("pos_x", "pos_y", "vel_x", "vel_y", "heading"):
                    d = float(np.abs(o.s[n] - traj[t][n]).max())
                    worst = max(worst, d)
                    if first_bad is None and not np.array_equal(o.s[n], traj[t][n]):
                        first_bad = (t, n, d)
                flags_ok = np.array_equal(o.s["flags"] & MASK, traj[t]["flags"])
                if first_bad is None and not flags_ok:
                    first_bad = (t, "flags", 0.0)
            return o, worst, first_bad
        finally:
            orc.set_libm()

    o_dev, worst_dev, first_dev = oracle_run(_GpuLibm())
    assert first_dev is None, "oracle on the device's libm bits leaves the GPU trajectory at (step, field, |diff|) = %r; " \
                              "worst %g" % (first_dev, worst_dev)
    assert np.array_equal(o_dev.s["flags"] & MASK, traj[-1]["flags"])
    # ... and the control: on glibc's bits the same oracle does NOT reproduce these episodes
    o_host, worst_host, first_host = oracle_run(None)
    assert first_host is not None and worst_host > 1e-3, (first_host, worst_host)
