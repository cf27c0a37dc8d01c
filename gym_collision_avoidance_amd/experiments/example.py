"""Minimum working example (reference: gym_collision_avoidance/experiments/src/example.py) -- BASELINE.json config 1:
a 4-agent swap scenario (fixture 4_agents_500_cases[0]), RVOPolicy + UnicycleDynamics, single env.

    GYM_CONFIG_CLASS=Example python -m gym_collision_avoidance_amd.experiments.example
"""
import os

os.environ.setdefault("GYM_CONFIG_CLASS", "Example")
import numpy as np  # noqa: E402

from gym_collision_avoidance_amd.envs import test_cases as tc  # noqa: E402
from gym_collision_avoidance_amd.envs.collision_avoidance_env import CollisionAvoidanceEnv  # noqa: E402


def main(num_steps=200, verbose=True):
    env = CollisionAvoidanceEnv()
    agents = tc.full_test_suite(4, 0, policies="RVO")       # agents 0/1 swap across the x axis, 2/3 antipodal
    env.set_agents(agents)
    obs, _ = env.reset()
    total = np.zeros(len(agents))
    for i in range(num_steps):
        actions = {}    # every policy is internal (RVO): nothing to supply, like env_utils.run_episode's step(None)
        obs, rewards, terminated, truncated, info = env.step(actions)
        total += rewards
        if terminated:
            if verbose:
                print("All agents finished after %d steps, rewards %s" % (i + 1, total))
            break
    return bool(terminated), [a.is_at_goal for a in env.agents]


def main_two_agents(num_steps=100, verbose=True):
    """The reference's own example (experiments/src/example.py:12-66): agent 0 runs an external (learning) policy fed
    the constant action [1.0, 0.5], agent 1 runs the pre-trained GA3C-CADRL network."""
    env = CollisionAvoidanceEnv()
    agents = tc.get_testcase_two_agents()
    [agent.policy.initialize_network() for agent in agents if hasattr(agent.policy, "initialize_network")]
    env.set_agents(agents)
    env.reset()
    terminated = False
    for i in range(num_steps):
        actions = {0: np.array([1.0, 0.5])}
        obs, rewards, terminated, truncated, which_agents_done = env.step(actions)
        if terminated:
            if verbose:
                print("All agents finished!")
            break
    return bool(terminated), [a.is_at_goal for a in env.agents], [a.in_collision for a in env.agents]


if __name__ == "__main__":
    main()
    main_two_agents()
    print("Experiment over.")
