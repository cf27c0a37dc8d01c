"""Scenario builders -> list[Agent] (reference: gym_collision_avoidance/envs/test_cases.py).

Covered: the registries (`policy_dict`, `sensor_dict`, `dynamics_dict`, test_cases.py:68-96), the fixture tables
(`preset_testCases(n, full_test_suite=True)`, :593-624 -- shipped as data/test_cases.npz, converted from the reference's
pickles by oracle/gen_golden.py), `cadrl_test_case_to_agents` (:495-590), `full_test_suite` (:593+ plumbing used by
run_full_test_suite), `get_testcase_two_agents` (:144-175), `gen_circle_test_case` (:900-911) and the hand-written
small presets, and random scenario generation (`get_testcase_random` -> scenario_generator.py, bit-identical to
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


def fixture_table(num_agents):
    """float64 [500, N, 6] = px, py, gx, gy, pref_speed, radius: the reference's {N}_agents_500_cases.p."""
    if num_agents not in _tables:
        with np.load(_DATA) as z:
            key = "n%d" % num_agents
            if key not in z:
                raise FileNotFoundError("no 500-case fixture for %d agents (have: %s)" % (num_agents, sorted(z.keys())))
            _tables[num_agents] = z[key]
    return _tables[num_agents]


def preset_testCases(num_agents, full_test_suite=False, vpref_constraint=False, radius_bounds=None, carrl=False,
                     seed=None):
    """list of [N,6] arrays.  full_test_suite=True -> the 500-case fixture; otherwise the small hand-written presets
    of test_cases.py:626-897 that are plain data (1, 2 and the asymmetric 3/4-agent cases)."""
    if full_test_suite:
        if vpref_constraint or carrl or seed is not None:
            raise NotImplementedError("only the plain {N}_agents_500_cases fixtures are shipped")
        return list(fixture_table(num_agents))
    a = np.array
    if num_agents == 1:
        return [a([[-3.0, 0.0, 3.0, 0.0, 1.0, 0.3]])]
    if num_agents == 2:
        return [a([[-3.0, 0.0, 3.0, 0.0, 1.0, 0.3], [3.0, 0.0, -3.0, 0.0, 1.0, 0.3]]),      # swap
                a([[-3.0, -1.5, 3.0, 1.5, 1.0, 0.5], [-3.0, 1.5, 3.0, -1.5, 1.0, 0.5]]),    # crossing
                a([[-2.0, -1.5, 2.0, 1.5, 1.0, 0.5], [-2.0, 1.5, 2.0, -1.5, 0.5, 0.5]])]
    raise NotImplementedError("hand-written presets for %d agents are not restated; use the fixtures" % num_agents)


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
