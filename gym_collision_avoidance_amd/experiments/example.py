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


if __name__ == "__main__":
    main()
    print("Experiment over.")
