"""Scenario builders -> list[Agent] (reference: gym_collision_avoidance/envs/test_cases.py).

Covered: the registries (`policy_dict`, `sensor_dict`, `dynamics_dict`, test_cases.py:68-96), the fixture tables
(`preset_testCases(n, full_test_suite=True)`, :593-624 -- shipped as data/test_cases.npz, converted from the reference's
pickles by oracle/gen_golden.py), `cadrl_test_case_to_agents` (:495-590), `full_test_suite` (:593+ plumbing used by
run_full_test_suite), `get_testcase_two_agents` (:144-175), `get_testcase_crazy`, `get_testcase_two_agents_laserscanners`,
`small_test_suite`, `formation`, `make_testcase_huge`, `yaml_to_agents`, `gen_circle_test_case` (:900-911), the hand-written
presets (data/presets.npz), and random scenario generation (`get_testcase_random` -> scenario_generator.py, bit-identical to
gen_rand_testcases.py under the same np.random seed).
"""
import os

import numpy as np

from gym_collision_avoidance_amd.envs import Config
from gym_collision_avoidance_amd.envs.agent import Agent
from gym_collision_avoidance_amd.envs.dynamics import (ExternalDynamics, UnicycleDynamics,
                                                       UnicycleDynamicsMaxTurnRate)
from gym_collision_avoidance_amd.envs.policies import (CARRLPolicy, ExternalPolicy, GA3CCADRLPolicy, LearningPolicy,
                                                       LearningPolicyGA3C, NonCooperativePolicy, RVOPolicy,
                                                       StaticPolicy)
from gym_collision_avoidance_amd.envs.sensors import LaserScanSensor, OtherAgentsStatesSensor

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "data", "test_cases.npz")

policy_dict = {
    "RVO": RVOPolicy,
    "GA3C_CADRL": GA3CCADRLPolicy,
    "noncoop": NonCooperativePolicy,
    "carrl": CARRLPolicy,
    "external": ExternalPolicy,
    "learning": LearningPolicy,
    "learning_ga3c": LearningPolicyGA3C,
    "static": StaticPolicy,
}
sensor_dict = {"other_agents_states": OtherAgentsStatesSensor, "laserscan": LaserScanSensor}
dynamics_dict = {"unicycle": UnicycleDynamics, "unicycle_max_turn_rate": UnicycleDynamicsMaxTurnRate,
                 "external": ExternalDynamics}

_tables = {}


def fixture_table(num_agents, carrl=False, seed=None):
    """float64 [500, N, 6] = px, py, gx, gy, pref_speed, radius: the reference's {N}_agents_500_cases.p; carrl / seed: its
    `_carrl` / `_carrl_seed00x` variants (test_cases.py:618-622; shipped for two agents, seeds 0 .. 4)."""
    key = "n%d" % num_agents + ("_carrl" if carrl else "") + ("_seed%03d" % seed if seed is not None else "")
    if key not in _tables:
        with np.load(_DATA) as z:
            if key not in z:  # (the reference fails the same way: open() of a pickle it does not ship)
                raise FileNotFoundError("no 500-case fixture %r (have: %s)" % (key, sorted(z.keys())))
            _tables[key] = z[key]
    return _tables[key]


_presets = None


def preset_testCases(num_agents, full_test_suite=False, vpref_constraint=False, radius_bounds=None, carrl=False,
                     seed=None):
    """list of [N,6] arrays.  full_test_suite=True -> the 500-case fixture (test_cases.py:593-624); otherwise the reference's
    hand-written presets for this agent count (test_cases.py:626-897: 1, 2, 3 / 4, 5, 6, 10 and 20 agents), shipped as
    data/presets.npz -- recorded from the imported reference by oracle/gen_presets.py, like the fixture tables."""
    global _presets
    if full_test_suite:
        if vpref_constraint:   # test_cases.py:603-607: a `vpref1.0_r<lo>-<hi>/` directory the reference does not ship either
            raise FileNotFoundError("the vpref_constraint fixtures (test_cases/vpref1.0_r%s-%s/) are not part of the "
                                    "reference repository" % tuple(radius_bounds or ("?", "?")))
        return list(fixture_table(num_agents, carrl=carrl, seed=seed))
    if _presets is None:
        with np.load(os.path.join(os.path.dirname(_DATA), "presets.npz")) as z:
            _presets = {k: z[k] for k in z.files}
    cases, i = [], 0
    while "n%d_%d" % (num_agents, i) in _presets:
        cases.append(_presets["n%d_%d" % (num_agents, i)].copy())
        i += 1
    if not cases:  # (the reference prints "invalid num_agents" and fails on the unbound list)
        raise ValueError("no hand-written presets for %d agents (the reference defines 1, 2, 3, 4, 5, 6, 10, 20)" % num_agents)
    return cases


def small_test_suite(num_agents, test_case_index, policies="learning", agents_dynamics="unicycle",
                     agents_sensors=("other_agents_states",), vpref_constraint=False, radius_bnds=None):
    """One of the hand-written presets as Agents (test_cases.py:313-330)."""
    return cadrl_test_case_to_agents(preset_testCases(num_agents)[test_case_index], policies=policies,
                                     agents_dynamics=agents_dynamics, agents_sensors=agents_sensors)


def gen_circle_test_case(num_agents, radius):
    """Agents evenly spaced on a circle, each heading to the antipode (test_cases.py:900-911)."""
    tc = np.zeros((num_agents, 6))
    for i in range(num_agents):
        th = 2 * np.pi * i / num_agents
        tc[i] = [radius * np.cos(th), radius * np.sin(th), radius * np.cos(th + np.pi), radius * np.sin(th + np.pi),
                 1.0, 0.5]
    return tc


def cadrl_test_case_to_agents(test_case, policies="RVO", policy_distr=None, agents_dynamics="unicycle",
                              agents_sensors=("other_agents_states",), policy_to_ensure=None, prev_agents=None):
    """[N,6] legacy CADRL rows -> Agents (test_cases.py:495-590).  `policies`: one name for everybody, or a list
    with one name per agent (policy_distr=None) / a pool sampled with probabilities policy_distr."""
    num_agents = np.shape(test_case)[0]
    if isinstance(policies, str):
        names = [policies] * num_agents
    elif isinstance(policies, (list, tuple)):
        if policy_distr is None:
            assert len(policies) >= num_agents
            names = list(policies)
        else:
            assert len(policies) == len(policy_distr)
            names = list(np.random.choice(policies, num_agents, p=policy_distr))
            if policy_to_ensure is not None and policy_to_ensure not in names:
                names[np.random.randint(len(names))] = policy_to_ensure
    else:
        raise NotImplementedError("policies must be a str or a list of str")
    sensors = [sensor_dict[s] for s in agents_sensors]
    agents = []
    for i, row in enumerate(test_case):
        px, py, gx, gy, pref_speed, radius = [float(v) for v in row[:6]]
        if Config.EVALUATE_MODE:
            heading = np.arctan2(gy - py, gx - px)  # toward the goal (:554-556)
        else:
            heading = np.random.uniform(-np.pi, np.pi)
        if prev_agents is not None and names[i] == prev_agents[i].policy.str:
            prev_agents[i].reset(px=px, py=py, gx=gx, gy=gy, pref_speed=pref_speed, radius=radius, heading=heading)
            agents.append(prev_agents[i])
        else:
            agents.append(Agent(px, py, gx, gy, radius, pref_speed, heading, policy_dict[names[i]],
                                dynamics_dict[agents_dynamics], sensors, i))
    return agents


def full_test_suite(num_agents, test_case_index, policies="RVO", agents_dynamics="unicycle",
                    agents_sensors=("other_agents_states",), prev_agents=None, **_ignored):
    """One case of the 500-case suite as Agents (reference test_cases.py `full_test_suite`)."""
    case = fixture_table(num_agents)[test_case_index]
    return cadrl_test_case_to_agents(case, policies=policies, agents_dynamics=agents_dynamics,
                                     agents_sensors=agents_sensors, prev_agents=prev_agents)


def get_testcase_two_agents(policies=("learning", "GA3C_CADRL")):
    """Two agents swapping corners (test_cases.py:144-175)."""
    g = 3
    return [Agent(-g, -g, g, g, 0.5, 1.0, 0.0, policy_dict[policies[0]], UnicycleDynamics,
                  [OtherAgentsStatesSensor], 0),
            Agent(g, g, -g, -g, 0.5, 1.0, np.pi, policy_dict[policies[1]], UnicycleDynamics,
                  [OtherAgentsStatesSensor], 1)]


def get_testcase_crazy(policy="GA3C_CADRL"):
    """Three agents in a corridor-like squeeze: the ego (`policy`) drives up the y-axis past two RVO agents, one with it,
    one against it (test_cases.py:99-141)."""
    rows = [(0.0, 0.0, 0.0, 8.0, np.pi / 2, policy), (-1.2, 0.0, -1.2, 5.0, np.pi / 2, "RVO"),
            (-1.2, 2.0, -1.2, -3.0, -np.pi / 2, "RVO")]
    return [Agent(px, py, gx, gy, 0.8, 1.0, hd, policy_dict[pol], UnicycleDynamics, [OtherAgentsStatesSensor], i)
            for i, (px, py, gx, gy, hd, pol) in enumerate(rows)]


def get_testcase_two_agents_laserscanners(policy="RVO"):
    """Two agents swapping corners that observe through LaserScanSensor only (test_cases.py:178-209).  The reference
    hard-codes its legacy PPOPolicy (out of scope here); `policy` names any registered policy."""
    g = 3
    return [Agent(-g, -g, g, g, 0.5, 1.0, 0.0, policy_dict[policy], UnicycleDynamics, [LaserScanSensor], 0),
            Agent(g, g, -g, -g, 0.5, 1.0, np.pi, policy_dict[policy], UnicycleDynamics, [LaserScanSensor], 1)]


# goal layouts of formation(): six goals per letter, in units of 2 m (test_cases.py:426-479)
_FORMATIONS = {
    "A": [(-1.5, 0.0), (1.5, 0.0), (0.75, 1.5), (-0.75, 1.5), (0.0, 1.5), (0.0, 3.0)],
    "C": [(0.0, 0.0), (-0.5, 1.0), (-0.5, 2.0), (0.0, 3.0), (1.5, 0.0), (1.5, 3.0)],
    "L": [(0.0, 0.0), (0.0, 1.0), (0.0, 2.0), (0.0, 3.0), (0.75, 0.0), (1.5, 0.0)],
    "D": [(0.0, 0.0), (0.0, 1.5), (0.0, 3.0), (1.5, 1.5), (1.2, 2.5), (1.2, 0.5)],
    "R": [(0.0, 0.0), (0.0, 1.5), (0.0, 3.0), (1.3, 2.8), (1.2, 1.7), (1.7, 0.0)],
}


def formation(agents, letter, num_agents=6):
    """Send the agents, from where they are, to the points of a letter (A, C, L, D, R) in a random assignment
    (test_cases.py:425-492): the next env.reset() starts them at their current positions with the new goals."""
    goals = 2.0 * np.array(_FORMATIONS[letter])
    order = np.arange(num_agents)
    np.random.shuffle(order)
    for agent in agents:
        px, py = agent.pos_global_frame
        gx, gy = goals[order[agent.id]]
        agent.reset(px=px, py=py, gx=gx, gy=gy, heading=agent.heading_global_frame)
    return agents


def make_testcase_huge(num_test_cases=1, num_agents=100, side_length=25, speed_bnds=[0.5, 2.0], radius_bnds=[0.2, 0.8],
                       policies="GA3C_CADRL"):
    """[num_test_cases, num_agents, 6] crowd scenarios (test_cases.py:914-976): per agent a speed and a radius, then a start
    at least 2 m (surface to surface) from every earlier START and a goal at least 2 m from every earlier GOAL and 5 m from
    its own start, all rejection-sampled from the square [-side_length, side_length]^2.  Same np.random draws, in the
    same order, as the reference.  (Up to 64 agents an env is one workgroup tile of the step kernels; 65 .. 1024 agents run the
    one-thread-per-agent kernel of csrc/cagpu_big.inc over CaOut.workspace -- DESIGN.md section 3c.)"""
    cases = np.empty((num_test_cases, num_agents, 6))
    for c in cases:
        for i in range(num_agents):
            speed = np.random.uniform(speed_bnds[0], speed_bnds[1])
            radius = np.random.uniform(radius_bnds[0], radius_bnds[1])

            def clearance(x, y, cx, cy):   # smallest surface distance to the points (cx, cy) of the agents placed so far
                if i == 0:
                    return np.inf
                return min(np.linalg.norm(np.array([x - o[cx], y - o[cy]])) - o[5] - radius for o in c[:i])
            gap = -np.inf
            while gap < 2.0:
                px, py = np.random.uniform(-side_length, side_length), np.random.uniform(-side_length, side_length)
                gap = clearance(px, py, 0, 1)
            gap, trip = -np.inf, -np.inf
            while gap < 2.0 or trip < 5.0:
                gx, gy = np.random.uniform(-side_length, side_length), np.random.uniform(-side_length, side_length)
                gap = clearance(gx, gy, 2, 3)
                trip = np.linalg.norm(np.array([px - gx, py - gy]))
            c[i] = [px, py, gx, gy, speed, radius]
    return cases


def get_testcase_huge(seed=None):
    """The reference's 100-agent scene (test_cases.py:979-992).  It unpickles `test_cases/100agents.p`, a file the
    reference repository does not ship (its own call raises FileNotFoundError), written by `make_testcase_huge` with
    its defaults -- so the scene is drawn here by that function (100 agents in a 50 m square; `seed`: np.random.seed
    first) and handed to GA3C-CADRL agents with unicycle dynamics like the reference's."""
    if seed is not None:
        np.random.seed(seed)
    return cadrl_test_case_to_agents(make_testcase_huge(1, 100, 25)[0], policies="GA3C_CADRL", agents_dynamics="unicycle",
                                     agents_sensors=["other_agents_states"])


def yaml_to_agents(agents_yaml):
    """[{name: {start_x, start_y, goal_x, goal_y, policy, dynamics}}, ...] -> Agents with radius 0.5, preferred speed 1,
    heading 0 (test_cases.py:1021-1041)."""
    agents = []
    for i, item in enumerate(agents_yaml):
        d = item[list(item.keys())[0]]
        agents.append(Agent(d["start_x"], d["start_y"], d["goal_x"], d["goal_y"], 0.5, 1.0, 0.0, policy_dict[d["policy"]],
                            dynamics_dict[d["dynamics"]], [OtherAgentsStatesSensor], i))
    return agents


def get_testcase_random(num_agents=None, side_length=4, speed_bnds=[0.5, 2.0], radius_bnds=[0.2, 0.8],
                        policies="learning", policy_distr=None, agents_dynamics="unicycle",
                        agents_sensors=["other_agents_states"], policy_to_ensure=None, prev_agents=None):
    """A random scenario (the reference's default TEST_CASE_FN, test_cases.py:212-253): 2 .. MAX_NUM_AGENTS agents,
    world size drawn from the {num_agents range: side_length range} list, 15 % swap / 15 % circle / 70 % rejection-
    sampled random starts and goals (scenario_generator.py).  Same np.random draws as the reference: under the same
    seed the scenario is the reference's, bit for bit."""
    from gym_collision_avoidance_amd.envs import scenario_generator
    if num_agents is None:
        num_agents = np.random.randint(2, Config.MAX_NUM_AGENTS_IN_ENVIRONMENT + 1)
    if type(side_length) is list:
        for comp in side_length:
            if comp["num_agents"][0] <= num_agents < comp["num_agents"][1]:
                side_length = np.random.uniform(comp["side_length"][0], comp["side_length"][1])
        assert type(side_length) == float
    case = scenario_generator.generate_rand_test_case_multi(num_agents, side_length, speed_bnds, radius_bnds)
    return cadrl_test_case_to_agents(case, policies=policies, policy_distr=policy_distr,
                                     agents_dynamics=agents_dynamics, agents_sensors=agents_sensors,
                                     policy_to_ensure=policy_to_ensure, prev_agents=prev_agents)
