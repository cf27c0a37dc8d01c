"""CPU-only tests: the C-ABI library exports what include/cagpu.h declares, the host-side mirror of the reference
interface (Config / Agent / registries / wrappers / sharding), and the multi-process (gloo) path.  No kernels run."""
import ctypes
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests import envtools
from tests import golden_util as gu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "cagpu.h")).read()
    declared = set(re.findall(r"\b(cagpu_[a-z_]+)\s*\(", hdr))
    assert {"cagpu_version", "cagpu_last_error", "cagpu_reset", "cagpu_step", "cagpu_rollout", "cagpu_orca",
            "cagpu_observe", "cagpu_step_map", "cagpu_laserscan"} <= declared
    so = os.path.join(REPO, "gym_collision_avoidance_amd", "libcagpu.so")
    if not os.path.exists(so):
        from gym_collision_avoidance_amd import build_native
        build_native.build()
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), name
    lib.cagpu_version.restype = ctypes.c_int
    from gym_collision_avoidance_amd import _native as nat
    assert lib.cagpu_version() == nat.ABI_VERSION == int(re.search(r"#define CAGPU_VERSION (\d+)", hdr).group(1))   # host-only call


def test_ctypes_structs_match_header_layout():
    from gym_collision_avoidance_amd import _native as nat
    assert ctypes.sizeof(nat.CaParams) == 8 * 4 + 17 * 8
    assert ctypes.sizeof(nat.CaState) == 24 * 8 and nat.CaState.turning_dir.offset == 20 * 8 and nat.CaState.ext_state.offset == 23 * 8 and ctypes.sizeof(nat.CaOut) == 8 * 8 and nat.CaOut.workspace.offset == 6 * 8
    assert nat.CaState.next_action.offset == 19 * 8 and nat.CaOut.orca_vel.offset == 5 * 8 and nat.CaParams.ragged.offset == 28
    assert ctypes.sizeof(nat.CaAutoReset) == 56 and nat.CaAutoReset.reset_obs.offset == 32
    assert nat.CaAutoReset.reset_plan.offset == 40 and nat.CaAutoReset.heading_seed.offset == 48
    assert nat.CaParams.dt.offset == 32
    assert ctypes.sizeof(nat.CaNet) == 16 * 8 and nat.CaNet.rows_scratch.offset == 12 * 8 and nat.CaNet.net_index.offset == 14 * 8 and nat.CaNet.packed.offset == 15 * 8
    assert ctypes.sizeof(nat.CaMap) == 8 + 2 * 4 + 3 * 8 and nat.CaMap.cell.offset == 16
    assert ctypes.sizeof(nat.CaScan) == 2 * 8 + 4 * 4 + 4 * 8 and nat.CaScan.min_angle.offset == 32


def test_map_indices_and_laserscan_sensor_registration():
    """Map.world_coordinates_to_map_indices (Map.py:26-32) and the 'laserscan' entry of sensor_dict"""
    from tests import envtools
    Config, tc, Env = envtools.fresh("Laser4")
    from gym_collision_avoidance_amd.envs.Map import Map
    m = Map(16, 16, 0.1)
    assert m.static_map.shape == (160, 160) and not m.static_map.any()
    (gx, gy), ok = m.world_coordinates_to_map_indices([0.0, 0.0])
    assert (gx, gy, ok) == (80, 80, True)
    (gx, gy), ok = m.world_coordinates_to_map_indices([-7.95, 7.95])
    assert (gx, gy, ok) == (0, 0, True)
    assert m.world_coordinates_to_map_indices([8.5, 0.0])[1] is False
    s = tc.sensor_dict["laserscan"]()
    assert (s.name, s.num_beams, s.num_to_store, s.max_range) == ("laserscan", 512, 3, 6)
    assert Config.STATE_INFO_DICT["laserscan"]["size"] == (3, 512)
    env = Env()
    assert env.observation_space.spaces[0].spaces["laserscan"].shape == (3, 512)
    env.set_static_map("world_maps/002.png")   # a path like the reference takes (read when the episode starts)
    assert env.static_map_filename == "world_maps/002.png"
    envtools.default()


def test_bad_arguments_are_loud_errors_not_crashes():
    from gym_collision_avoidance_amd import _native as nat, core
    lib = nat.lib()
    assert lib.cagpu_step(None, None, None, None, None, None) == nat.CA_EINVAL
    assert b"NULL" in lib.cagpu_last_error()
    p = core.make_params(4, 1300)    # beyond the large-env kernel's one thread per agent (1024 = the largest workgroup)
    s, o = nat.CaState(), nat.CaOut()
    assert lib.cagpu_observe(ctypes.byref(p), ctypes.byref(s), ctypes.byref(o), None) == nat.CA_EUNSUPPORTED
    p = core.make_params(4, 70)      # more than 64 agents: the large-env kernel wants its workspace
    assert lib.cagpu_observe(ctypes.byref(p), ctypes.byref(s), ctypes.byref(o), None) == nat.CA_EINVAL
    assert b"workspace" in lib.cagpu_last_error()
    assert lib.cagpu_workspace_bytes(ctypes.byref(core.make_params(4, 64))) == 0
    assert lib.cagpu_workspace_bytes(ctypes.byref(p)) % (256 * 70 * 60) == 0
    with pytest.raises(nat.CagpuError):
        nat.check(-2)


def test_no_cpu_fallback_in_product_path():
    import torch
    from gym_collision_avoidance_amd import _native as nat, core
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nat.CagpuError):
        core.BatchedSim(core.make_params(2, 3))
    for root, _, files in os.walk(os.path.join(REPO, "gym_collision_avoidance_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_config_singleton_and_env_var_selection():
    Config, tc, Env = envtools.fresh("Clip6")
    assert Config.MAX_NUM_AGENTS_IN_ENVIRONMENT == 6 and Config.MAX_NUM_OTHER_AGENTS_OBSERVED == 3
    assert Config.AGENT_SORTING_METHOD == "closest_last" and Config.DT == 0.1 and Config.EVALUATE_MODE
    assert Config.STATE_INFO_DICT["other_agents_states"]["size"] == (3, 7)
    envtools.default()
    from gym_collision_avoidance_amd.envs import Config as C0
    assert C0.DT == 0.2 and C0.MAX_NUM_AGENTS_IN_ENVIRONMENT == 4 and C0.MAX_TIME_RATIO == 2.0
    assert C0.STATES_IN_OBS[0] == "is_learning" and C0.RVO_TIME_HORIZON == 5.0


def test_agent_initial_condition_matches_reference_reset():
    """Agent.reset arithmetic (agent.py:59-138) against the t=0 record of the reference"""
    Config, tc, Env = envtools.fresh("Bench10")
    meta, eps = gu.load("rvo10")
    ep = eps[0]
    agents = tc.full_test_suite(10, 0, policies="RVO")
    for i, a in enumerate(agents):
        assert np.allclose(a.pos_global_frame, [ep.col(0, "pos_x")[i], ep.col(0, "pos_y")[i]], atol=0)
        assert abs(a.heading_global_frame - ep.col(0, "heading")[i]) < 1e-15
        assert abs(a.time_remaining_to_reach_goal - ep.col(0, "time_remaining")[i]) < 1e-12
        assert abs(a.straight_line_time_to_reach_goal - ep.col(0, "slt")[i]) < 1e-12
        assert abs(a.dist_to_goal - ep.obs[0][i, 2]) < 1e-12 and abs(a.heading_ego_frame - ep.obs[0][i, 3]) < 1e-12
        assert not a.is_done and a.t == 0.0 and a.policy.str == "RVO" and not a.policy.is_external


def test_registries_and_plugin_flags():
    Config, tc, Env = envtools.fresh("Swap4")
    from gym_collision_avoidance_amd import _native as nat
    assert set(tc.policy_dict) >= {"RVO", "noncoop", "static", "external", "learning", "learning_ga3c"}
    ids = {k: tc.policy_dict[k].kernel_id for k in ("RVO", "noncoop", "static", "external", "learning", "learning_ga3c")}
    assert ids == {"RVO": nat.POL_RVO, "noncoop": nat.POL_NONCOOP, "static": nat.POL_STATIC,
                   "external": nat.POL_EXTERNAL, "learning": nat.POL_LEARNING, "learning_ga3c": nat.POL_LEARNING_GA3C}
    lp = tc.policy_dict["learning"]()
    assert lp.is_external and lp.is_still_learning and lp.str == "learning"
    ag = tc.get_testcase_two_agents()
    act = lp.external_action_to_action(ag[0], np.array([0.5, 1.0]))
    assert np.allclose(act, [0.5, math.pi / 3])
    g = tc.policy_dict["learning_ga3c"]()
    assert np.allclose(g.external_action_to_action(ag[0], 7), [0.5, math.pi / 6])
    assert tc.fixture_table(10).shape == (500, 10, 6) and len(tc.preset_testCases(4, full_test_suite=True)) == 500
    s = tc.sensor_dict["other_agents_states"]()
    s.set_args({"max_num_other_agents_observed": 2, "agent_sorting_method": "closest_last"})
    assert s.max_num_other_agents_observed == 2 and s.name == "other_agents_states"
    np.random.seed(3)
    rnd = tc.get_testcase_random(num_agents=3)
    assert len(rnd) == 3 and all(a.policy.str == "learning" for a in rnd)


def test_env_constructs_without_gpu_and_checks_arguments():
    Config, tc, Env = envtools.fresh("Swap4")
    env = Env()
    assert env.num_agents == 4 and env.dt_nominal == 0.1 and env.episode_step_number is None
    assert env.action_space.low.tolist() == [0.0, -math.pi / 3] and env.max_possible_reward == 1.0
    assert set(env.observation[0]) == set(Config.STATES_IN_OBS)
    assert env.observation_space.spaces[3]["other_agents_states"].shape == (3, 7)
    with pytest.raises(RuntimeError):
        env.step({})
    with pytest.raises(AssertionError):
        env.set_testcase("no_such_fn", {})


def test_array_wrapper_layout():
    Config, tc, Env = envtools.fresh("Bench10")
    from gym_collision_avoidance_amd.envs.wrappers import (MultiagentDictToMultiagentArrayWrapper,
                                                           MultiagentFlattenDictWrapper)
    env = Env()
    w = MultiagentDictToMultiagentArrayWrapper(env, Config.STATES_IN_OBS, 10)
    assert w.obs_shape == (10, 69)
    assert w.observation_indices[3]["other_agents_states"] == [6, 69] and w.observation_indices[0]["radius"] == [5, 6]
    arr = w.observation(env.observation)
    assert arr.shape == (10, 69) and not arr.any()
    f = MultiagentFlattenDictWrapper(env, Config.STATES_IN_OBS, 10)
    assert f.obs_shape == (690,) and f.observation_indices[1]["BOUNDS"] == [69, 138]


def test_shard_env_ids():
    from gym_collision_avoidance_amd.sharding import shard_env_ids
    assert shard_env_ids(0, 1, 4096) == (0, 4096)
    assert shard_env_ids(3, 8, 4096) == (3 * 4096, 8 * 4096)
    with pytest.raises(ValueError):
        shard_env_ids(8, 8, 1)
    # shards replay disjoint case streams, exactly as one big batch would
    seen = set()
    for r in range(4):
        off, stride = shard_env_ids(r, 4, 5)
        for e in range(5):
            for k in range(3):
                seen.add((off + e + k * stride))
    assert seen == set(range(60))


_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from gym_collision_avoidance_amd.sharding import shard_env_ids, reduce_episode_stats, gather_episode_stats
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
off, stride = shard_env_ids(rank, world, 16)
stats = torch.arange(8, dtype=torch.float64) * (rank + 1)
tot = reduce_episode_stats(stats)
allg = gather_episode_stats(stats)
assert torch.equal(tot, torch.arange(8, dtype=torch.float64) * 3), tot
assert allg.shape == (2, 8) and torch.equal(allg[1], torch.arange(8, dtype=torch.float64) * 2)
assert torch.equal(stats, torch.arange(8, dtype=torch.float64) * (rank + 1))   # input untouched
assert (off, stride) == (16 * rank, 32)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_process_stats_reduction_gloo(tmp_path):
    """the N>1 path on CPU: one process per shard, gloo backend, world_size 2"""
    script = tmp_path / "w.py"
    script.write_text(_WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]


def test_random_scenarios_match_the_reference_bit_for_bit():
    """scenario_generator.py / test_cases.get_testcase_random against arrays recorded from the unmodified reference
    under the same np.random seeds (tests/golden/rand_cases.npz, oracle/gen_golden.py:rand_cases): swap / circle /
    rejection-sampled families, the env's default reset path with a random agent count, the side-length table, the
    policy lottery and the random initial headings of training mode"""
    import json
    from tests import envtools
    z = np.load(os.path.join(REPO, "tests", "golden", "rand_cases.npz"))
    Config, tc, Env = envtools.fresh("Train5")
    from gym_collision_avoidance_amd.envs import scenario_generator as sg
    kinds = set()
    for seed in range(60):
        n, side = 2 + seed % 9, 4.0 + (seed % 5)
        np.random.seed(seed)
        dice = np.random.rand()
        kinds.add("swap" if dice < 0.15 else "circle" if dice < 0.3 else "rand")
        np.random.seed(seed)
        got = sg.generate_rand_test_case_multi(n, side, [0.5, 2.0], [0.2, 0.8])
        assert np.array_equal(got, z["multi_%d" % seed]), seed
    assert kinds == {"swap", "circle", "rand"}
    for seed in range(12):     # the static-obstacle family (is_static=True -> generate_static_case)
        n, side = 2 + seed % 7, 3.0 + (seed % 4)
        np.random.seed(500 + seed)
        got = sg.generate_rand_test_case_multi(n, side, [0.5, 2.0], [0.2, 0.8], is_static=True)
        assert np.array_equal(got, z["static_%d" % seed]), seed
        assert n < 2 or (np.array_equal(got[1:, 0:2], got[1:, 2:4]) and got[0, 0] <= -1.5 <= 1.5 <= got[0, 2])
    assert int(z["max_agents"]) == Config.MAX_NUM_AGENTS_IN_ENVIRONMENT
    ref_args = json.loads(str(z["test_case_args"]))
    assert ref_args["policies"] == Config.TEST_CASE_ARGS["policies"] and \
        ref_args["policy_distr"] == Config.TEST_CASE_ARGS["policy_distr"]
    for seed in range(20):
        np.random.seed(1000 + seed)
        agents = tc.get_testcase_random(**Config.TEST_CASE_ARGS)
        want = z["env_%d" % seed]
        assert len(agents) == len(want)
        got = np.array([list(a._case_row()[0][:4]) + [a._case_row()[0][4], a._case_row()[0][5], a._case_row()[1]]
                        for a in agents])
        assert np.array_equal(got, want), seed
        assert [type(a.policy).__name__ for a in agents] == list(z["envpol_%d" % seed])
    # the env's default reset path uses it (collision_avoidance_env.py:345-362): no agents set -> a random scenario
    env = Env()
    assert env.test_case_fn is tc.get_testcase_random
    envtools.default()


def host_cases_from_philox(seed, num_cases, n, side, speed=(0.5, 2.0), radius=(0.2, 0.8), num_agents=None):
    """the HOST generator (bit-identical to the reference under np.random) driven by the device generator's uniform
    stream (oracle/philox_ref.py): what cagpu_generate_cases must return.  -> (cases [C, n, 6], family names).
    `num_agents=(lo, hi)`: the ragged form (cagpu_generate_cases_ragged) -- the count first, then the side length from
    the reference's list of range dicts (test_cases.py:224-241), rows past the count zero"""
    from oracle.philox_ref import PhiloxStream
    from gym_collision_avoidance_amd.envs import scenario_generator as sg

    class _NP(object):  # what scenario_generator reads from numpy, with `random` swapped for the Philox stream
        def __getattr__(self, name):
            return getattr(np, name)

    def preamble(st):  # the draws before the family dice
        k = n
        if num_agents is not None:
            k = min(num_agents[0] + int(st.rand() * (num_agents[1] - num_agents[0] + 1)), num_agents[1])
        s = side
        if isinstance(side, list):
            for comp in side:
                if comp["num_agents"][0] <= k < comp["num_agents"][1]:
                    s = comp["side_length"][0] + (comp["side_length"][1] - comp["side_length"][0]) * st.rand()
        elif not np.isscalar(side):
            s = side[0] + (side[1] - side[0]) * st.rand()
        return k, s
    out, kinds = [], []
    real = sg.np
    try:
        for c in range(num_cases):
            st = PhiloxStream(seed, c)
            shim = _NP()
            shim.random = st
            sg.np = shim
            k, s = preamble(st)
            dice = PhiloxStream(seed, c)
            preamble(dice)
            d = dice.rand()
            kinds.append("swap" if d < 0.15 else "circle" if d < 0.3 else "rand")
            rows = np.zeros((n, 6))
            rows[:k] = sg.generate_rand_test_case_multi(k, s, list(speed), list(radius))
            out.append(rows)
    finally:
        sg.np = real
    return np.array(out), kinds


def test_philox_stream_and_host_generator_under_it():
    """Philox4x32-10 known answers (Random123 kat_vectors), and the host generator driven by the Philox stream still
    honours the reference's acceptance rules"""
    from oracle.philox_ref import philox4x32_10, PhiloxStream
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    a, b = PhiloxStream(7, 3), PhiloxStream(7, 3)
    x = [a.rand() for _ in range(5)]
    assert x == list(b.rand(5)) and all(0.0 <= v < 1.0 for v in x)
    assert PhiloxStream(7, 4).rand() != x[0] and PhiloxStream(8, 3).rand() != x[0]
    cases, kinds = host_cases_from_philox(11, 40, 6, 4.0)
    assert set(kinds) == {"swap", "circle", "rand"}
    for cs in cases:
        for i in range(6):
            for j in range(i):
                clr = cs[i, 5] + cs[j, 5] + 0.2
                assert np.hypot(*(cs[i, 0:2] - cs[j, 0:2])) >= clr and np.hypot(*(cs[i, 2:4] - cs[j, 2:4])) >= clr
        assert np.all((cs[:, 4] >= 0.5) & (cs[:, 4] <= 2.0) & (cs[:, 5] >= 0.2) & (cs[:, 5] <= 0.8))


def test_map_image_loading(tmp_path):
    """Map(map_filename=...) like the reference's loader (Map.py:14-24): dark pixels are obstacles, images of another
    size are resized to the grid with nearest-neighbour sampling"""
    from PIL import Image
    from gym_collision_avoidance_amd.envs.Map import Map
    img = np.full((160, 160), 255, dtype=np.uint8)
    img[40:44, 30:130] = 0
    f = str(tmp_path / "world.png")
    Image.fromarray(img).save(f)
    m = Map(16, 16, 0.1, map_filename=f)
    assert m.static_map.shape == (160, 160) and m.static_map.dtype == bool
    assert m.static_map[40:44, 30:130].all() and m.static_map.sum() == 4 * 100
    big = np.kron(img, np.ones((2, 2), dtype=np.uint8))  # 320 x 320
    f2 = str(tmp_path / "world2.png")
    Image.fromarray(big).save(f2)
    assert np.array_equal(Map(16, 16, 0.1, map_filename=f2).static_map, m.static_map)
    # an 8-bit map whose brightest level is below 255, resized: scipy's bytescale leaves uint8 data alone, so the
    # reference's imresize never stretches it and np.invert(...).astype(bool) marks EVERY pixel != 255 as an obstacle
    dim = np.kron(np.where(img == 255, 200, 0).astype(np.uint8), np.ones((2, 2), dtype=np.uint8))
    f3 = str(tmp_path / "world3.png")
    Image.fromarray(dim).save(f3)
    assert Map(16, 16, 0.1, map_filename=f3).static_map.all()


def test_scenario_builders_match_the_reference():
    """make_testcase_huge / formation / get_testcase_crazy / the hand-written presets against what the imported reference
    returned under the same np.random seed (oracle/gen_presets.py -> tests/golden/builders.npz, data/presets.npz)"""
    Config, tc, _ = envtools.fresh("Bench10")    # (an EVALUATE_MODE config, like the one the vectors were recorded under)
    g = np.load(os.path.join(REPO, "tests", "golden", "builders.npz"))
    np.random.seed(3)
    assert np.array_equal(tc.make_testcase_huge(1, 12, 10), g["huge_seed3_12_10"])
    np.random.seed(0)
    agents = tc.cadrl_test_case_to_agents(tc.preset_testCases(6)[0], policies="noncoop")
    tc.formation(agents, "C")
    got = np.array([[*a.pos_global_frame, *a.goal_global_frame, a.heading_global_frame] for a in agents])
    np.testing.assert_allclose(got, g["formation_C_seed0"], rtol=0, atol=1e-12)
    crazy = tc.get_testcase_crazy("noncoop")
    got = np.array([[*a.pos_global_frame, *a.goal_global_frame, a.pref_speed, a.radius, a.heading_global_frame] for a in crazy])
    np.testing.assert_allclose(got, g["crazy"], rtol=0, atol=1e-12)
    assert [len(tc.preset_testCases(n)) for n in (1, 2, 3, 4, 5, 6, 10, 20)] == [2, 8, 9, 9, 2, 4, 1, 1]
    assert len(tc.small_test_suite(4, 7, policies="RVO")) == 4
    with pytest.raises(AssertionError):     # (like the reference's sensor: LaserScanSensor needs Config.USE_STATIC_MAP)
        tc.get_testcase_two_agents_laserscanners()
    a = tc.yaml_to_agents([{"robot": dict(start_x=0, start_y=1, goal_x=2, goal_y=3, policy="RVO", dynamics="unicycle")}])
    assert len(a) == 1 and a[0].radius == 0.5 and a[0].pref_speed == 1.0
    with pytest.raises(ValueError):
        tc.preset_testCases(7)


def test_plugin_descriptors_of_user_classes_and_sensor_argument_groups():
    """host logic of the round-4 plugin surface, without a device: a user Dynamics subclass becomes an ExternalDynamics slot
    whose agent is also queried on the host; per-agent sensor arguments are grouped into the pair most agents use (CaParams)
    and the further pairs (one cagpu_observe launch each); the mutable agent stand-in a user Dynamics.step writes to keeps
    float64 arithmetic under numpy >= 2"""
    Config, tc, Env = envtools.fresh("Bench10")
    from gym_collision_avoidance_amd import _native as nat
    from gym_collision_avoidance_amd.envs import collision_avoidance_env as cae
    from gym_collision_avoidance_amd.envs.agent import Agent
    from gym_collision_avoidance_amd.envs.dynamics import Dynamics, UnicycleDynamics
    from gym_collision_avoidance_amd.envs.policies import InternalPolicy, NonCooperativePolicy, RVOPolicy, StaticPolicy
    from gym_collision_avoidance_amd.envs.sensors import OtherAgentsStatesSensor

    class MyDyn(Dynamics):
        def step(self, action, dt):
            self.agent.heading_global_frame = action[1] + self.agent.heading_global_frame

    class MyPol(InternalPolicy):
        def find_next_action(self, obs, agents, i):
            return np.zeros(2)

    mk = lambda i, pol, dyn: Agent(float(i), 0.0, 5.0, float(i), 0.3, 1.0, None, pol, dyn, [OtherAgentsStatesSensor], i)
    agents = [mk(0, RVOPolicy, UnicycleDynamics), mk(1, RVOPolicy, MyDyn), mk(2, MyPol, UnicycleDynamics),
              mk(3, StaticPolicy, MyDyn), mk(4, NonCooperativePolicy, UnicycleDynamics)]
    env = Env()
    pol, dyn, isl, stl = env._plugin_ids(agents)
    assert pol == [nat.POL_RVO, nat.POL_EXTERNAL, nat.POL_EXTERNAL, nat.POL_STATIC, nat.POL_NONCOOP]
    assert dyn == [nat.DYN_UNICYCLE, nat.DYN_EXTERNAL, nat.DYN_UNICYCLE, nat.DYN_EXTERNAL, nat.DYN_UNICYCLE]
    assert env._host_policies == [1, 2] and env._host_dynamics == [1, 3]
    # sensor arguments: three agents on the default pair, two on (3, closest_last)
    for a in agents[3:]:
        a.sensors[0].set_args({"agent_sorting_method": "closest_last", "max_num_other_agents_observed": 3})
    K, clip, sort, others = env._sensor_args([agents])
    assert (K, clip, sort) == (Config.MAX_NUM_OTHER_AGENTS_OBSERVED, Config.MAX_NUM_OTHER_AGENTS_OBSERVED, nat.SORT_CLOSEST_FIRST)
    assert others == {(3, nat.SORT_CLOSEST_LAST): [(0, 3), (0, 4)]}
    agents[0].sensors[0].set_args({"agent_sorting_method": "by_colour"})
    with pytest.raises(ValueError):
        env._sensor_args([agents])
    # the stand-in: plain mutable attributes, np.float64 scalars, everything else read through
    proxy = cae._HostAgent(agents[1])
    MyDyn(proxy).step(np.array([1.0, 0.1], dtype=np.float32), 0.1)
    assert isinstance(proxy.heading_global_frame, np.float64)
    assert abs(float(proxy.heading_global_frame) - (float(np.float32(0.1)) + agents[1].heading_global_frame)) < 1e-15
    proxy.pos_global_frame += 1.0
    assert np.allclose(agents[1].pos_global_frame, [1.0, 0.0]) and np.allclose(proxy.pos_global_frame, [2.0, 1.0])
    assert proxy.pref_speed == agents[1].pref_speed and proxy.id == 1


def test_install_as_aliases_the_package_under_the_reference_name(tmp_path):
    """gym_collision_avoidance_amd.install_as(): the reference's import lines resolve to THIS package's module objects
    (one Config singleton), `experiments.src.*` maps to `experiments.*`, unknown names fail like any missing module, the
    gym stand-in is lazy (importing it does not instantiate Config) and an existing package of that name is not shadowed.
    Runs in a child process: the alias is a process-wide import hook."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import gym_collision_avoidance_amd as pkg
assert pkg.install_as(provide_gym=True) == "gym_collision_avoidance"
pkg.install_as()                                     # idempotent
import gym
gym.logger.set_level(40)
assert not any(m.startswith("gym_collision_avoidance_amd.envs") for m in sys.modules)
os.environ["GYM_CONFIG_CLASS"] = "Example"           # (the caller selects the Config class before its first import)
from gym_collision_avoidance.envs import Config
from gym_collision_avoidance.envs import test_cases as tc
import gym_collision_avoidance_amd.envs as E, gym_collision_avoidance_amd.envs.test_cases as TC
assert Config is E.Config and tc is TC and type(Config).__name__ == "Example"
from gym_collision_avoidance.experiments.src.env_utils import run_episode
import gym_collision_avoidance_amd.experiments.env_utils as EU
assert run_episode is EU.run_episode
from gym_collision_avoidance.envs.policies.RVOPolicy import RVOPolicy
from gym_collision_avoidance.envs.agent import Agent
from gym_collision_avoidance_amd.envs.agent import Agent as A2
assert Agent is A2 and gym.spaces.Box is E.spaces.Box
try:
    import gym_collision_avoidance.no_such_module
    raise SystemExit("missing module imported")
except ModuleNotFoundError:
    pass
os.makedirs(os.path.join(%r, "other_pkg"))
open(os.path.join(%r, "other_pkg", "__init__.py"), "w").close()
sys.path.insert(0, %r)
try:
    pkg.install_as("other_pkg")
    raise SystemExit("shadowed an importable package")
except ImportError:
    pass
print("alias ok")
''' % (REPO, str(tmp_path), str(tmp_path), str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("GYM_CONFIG_CLASS", "GYM_CONFIG_PATH")}
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and b"alias ok" in r.stdout, r.stdout.decode()[-2000:]


def test_carrl_fixture_family_and_huge_testcase():
    """preset_testCases(2, full_test_suite=True, carrl=True[, seed=k]) serves the reference's `_carrl` / `_carrl_seed00k` pickles
    (test_cases.py:618-622; shipped as data like the plain suites); a fixture the reference does not ship fails the same
    way its open() would; get_testcase_huge builds the 100-agent scene make_testcase_huge draws with its defaults"""
    Config, tc, Env = envtools.fresh("Bench10")
    plain = tc.preset_testCases(2, full_test_suite=True)
    carrl = tc.preset_testCases(2, full_test_suite=True, carrl=True)
    seeds = [tc.preset_testCases(2, full_test_suite=True, carrl=True, seed=k) for k in range(5)]
    assert len(plain) == len(carrl) == 500 and all(len(s_) == 500 and s_[0].shape == (2, 6) for s_ in seeds)
    assert not np.array_equal(np.array(plain), np.array(carrl)) and not np.array_equal(np.array(seeds[0]), np.array(seeds[1]))
    if os.path.isdir("/root/reference"):   # (build container: the data equal the pickles byte for byte)
        import pickle
        d = "/root/reference/gym_collision_avoidance/envs/test_cases/"
        for name, got in (("2_agents_500_cases_carrl.p", carrl), ("2_agents_500_cases_carrl_seed003.p", seeds[3])):
            with open(d + name, "rb") as f:
                want = pickle.load(f, encoding="latin1")
            assert all(np.array_equal(np.asarray(w_, np.float64), g_) for w_, g_ in zip(want, got))
    with pytest.raises(FileNotFoundError):
        tc.preset_testCases(3, full_test_suite=True, carrl=True)
    with pytest.raises(FileNotFoundError):
        tc.preset_testCases(2, full_test_suite=True, vpref_constraint=True, radius_bounds=[0.2, 0.8])
    agents = tc.get_testcase_huge(seed=5)
    np.random.seed(5)
    want = tc.make_testcase_huge(1, 100, 25)[0]
    assert len(agents) == 100 and type(agents[0].policy).__name__ == "GA3CCADRLPolicy"
    assert np.allclose([a._case_row()[0][:6] for a in agents], want)
    envtools.default()


def test_ga3c_gate_micro_op_orders_are_schedules_of_the_cell_update():
    """csrc/cagpu_ga3c.inc places the LSTM's cell update by hand: GATE_ORDER / GATE_ORDER_SEL list the single-instruction
    micro-ops (4 step + element) in the order they are issued behind the MFMAs.  Whatever the order, it must be a
    permutation that respects the update's data flow (gate_op: exp -> add -> rcp per gate, the o gate's reciprocal taken
    together with tanh(c)'s; K tanh(j), f * c', the cell fma, 2^c', the common denominator and its reciprocal, 1 - 2^c', h;
    the two selects last), with every consumer at least two positions behind its producers (a dependent instruction
    right behind a transcendental stalls the wave)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym_collision_avoidance_amd", "csrc",
                            "cagpu_ga3c.inc")).read()
    n_ops = {k: int(v) for k, v in re.findall(r"constexpr int (GATE_OPS(?:_SEL)?) = (\d+);", src)}
    assert n_ops == {"GATE_OPS": 80, "GATE_OPS_SEL": 88}
    for name, steps in (("GATE_ORDER", 20), ("GATE_ORDER_SEL", 22)):
        body = re.search(r"constexpr int %s\[GATE_OPS(?:_SEL)?\] = \{([^}]*)\};" % name, src).group(1)
        order = [int(x) for x in body.replace("\n", " ").split(",") if x.strip()]
        assert sorted(order) == list(range(4 * steps)), name
        pos = {op: i for i, op in enumerate(order)}
        deps = {}
        for r in range(4):
            op = lambda s: 4 * s + r  # noqa: E731
            for s in range(4, 11):
                deps[op(s)] = [op(s - 4)]
            deps[op(11)], deps[op(12)] = [op(9)], [op(10)]
            deps[op(13)] = [op(8), op(11), op(12)]
            deps[op(14)], deps[op(15)] = [op(13)], [op(14)]
            deps[op(16)], deps[op(17)] = [op(15), op(7)], [op(16)]
            deps[op(18)], deps[op(19)] = [op(14)], [op(18), op(17)]
            if steps > 20:
                deps[op(20)], deps[op(21)] = [op(13), op(12)], [op(19)]
        for o, ds in deps.items():
            for d in ds:
                assert pos[o] - pos[d] >= 2, (name, o, d, pos[o], pos[d])


def test_bench_lines_carry_the_provenance_of_the_library_they_ran_on():
    """round 6 (a profile must not go stale unnoticed): build_native writes the commit, the dirty
    flag and the digests of the kernel sources and of libcagpu.so beside the library; bench.provenance() -- stamped into every
    bench line and every profiles/r06_* record -- hashes the file that is actually loaded and only vouches for the build record
    when it belongs to that file"""
    import hashlib
    import bench
    from gym_collision_avoidance_amd import _native as nat, build_native as bn
    assert os.path.exists(bn.OUT), "build first (python -c 'import __graft_entry__ as g; g.build()')"
    info = bn.build_info()
    want = hashlib.sha256(open(bn.OUT, "rb").read()).hexdigest()
    assert info["lib_sha256"] == want
    pv = bench.provenance()
    assert pv["lib_sha256"] == hashlib.sha256(open(nat.LIB_PATH, "rb").read()).hexdigest() and len(pv["bench_py_sha256"]) == 64
    if os.path.abspath(nat.LIB_PATH) == os.path.abspath(bn.OUT) and not info["stale"]:
        assert pv["source_sha256"] == bn.source_digest() or info.get("git_dirty") is not None   # (sources edited after the build: the digest says so)
        assert "git_sha" in pv and "git_dirty" in pv
    else:
        assert "note" in pv and "git_sha" not in pv
    # the committed round-6 record names ONE library, and every line filed with it carries that library's hash
    import glob
    import json
    prov = json.load(open(os.path.join(REPO, "profiles", "r06_provenance.json")))
    for f in glob.glob(os.path.join(REPO, "profiles", "r06_*.json")):
        d = json.load(open(f))
        for rec in (d if isinstance(d, list) else [d]):
            p = rec.get("provenance") if isinstance(rec, dict) else None
            if p:
                assert p["lib_sha256"] == prov["lib_sha256"], f
