"""The 500-case evaluation suite, all cases at once (reference: experiments/src/run_full_test_suite.py:54-130, which
loops `reset_env` / `run_episode` over test cases x policies x agent counts, one Python episode at a time).

Here every test case is its own env of one batch: `run_suite` builds the agents exactly like the reference's
`reset_env` (test_cases.full_test_suite + policy.initialize_network + sensor.set_args), steps the batch until every env
is over and returns one row per case with `run_episode`'s statistics schema (env_utils.py:56-87).

    GYM_CONFIG_CLASS=FullTestSuite python -m gym_collision_avoidance_amd.experiments.run_full_test_suite
"""
import os

os.environ.setdefault("GYM_CONFIG_CLASS", "FullTestSuite")
import numpy as np  # noqa: E402

from gym_collision_avoidance_amd import _native as nat  # noqa: E402
from gym_collision_avoidance_amd.envs import Config  # noqa: E402
from gym_collision_avoidance_amd.envs import test_cases as tc  # noqa: E402
from gym_collision_avoidance_amd.envs.collision_avoidance_env import CollisionAvoidanceEnv  # noqa: E402
from gym_collision_avoidance_amd.experiments.env_utils import policies  # noqa: E402


def run_suite(policy="RVO", num_agents=4, test_cases=None, device="cuda:0", max_steps=2000):
    """-> pandas.DataFrame, one row per test case: num_agents, policy, test_case, total_reward [N], steps,
    time_to_goal [N], total_time_to_goal, extra_time_to_goal [N], collision, all_at_goal, any_stuck, outcome."""
    import pandas as pd
    import torch
    spec = policies[policy]
    cases = list(range(len(tc.fixture_table(num_agents)))) if test_cases is None else list(test_cases)
    per_env = []
    for c in cases:
        agents = tc.full_test_suite(num_agents, c, policies=spec["policy"],
                                    agents_sensors=spec.get("sensors", ["other_agents_states"]))
        for a in agents:
            if "checkpt_name" in spec:
                a.policy.initialize_network(**spec)
            for s in a.sensors:
                if "sensor_args" in spec:
                    s.set_args(spec["sensor_args"])
        per_env.append(agents)
    env = CollisionAvoidanceEnv(num_envs=len(cases), device=device)
    env.set_agents(per_env if len(cases) > 1 else per_env[0])
    env.reset()
    sim = env._sim
    # run_episode stops an episode at game_over (env_utils.py:45-52); the batch keeps stepping until its slowest env is
    # over, and the kernel keeps paying step rewards to the agents of finished envs (e.g. a timed-out agent standing
    # within GETTING_CLOSE_RANGE of another one), so every per-episode quantity is LATCHED at the step its env finishes.
    latched = ("t", "slt", "ep_reward", "flags")
    finish = torch.full((len(cases),), -1, dtype=torch.int32, device=sim.device)
    keep = {k: sim.state[k].clone() for k in latched}
    for t in range(1, max_steps + 1):
        sim.step()
        over = sim.game_over.bool()
        newly = (finish < 0) & over
        finish = torch.where(newly, torch.full_like(finish, t), finish)
        for k in latched:
            keep[k] = torch.where(newly[:, None], sim.state[k], keep[k])
        if t % 64 == 0 and bool((finish >= 0).all()):
            break
    for k in latched:   # envs that never finished within max_steps report their last state
        keep[k] = torch.where((finish < 0)[:, None], sim.state[k], keep[k])
    finish = torch.where(finish < 0, torch.full_like(finish, t), finish).cpu().numpy()
    st = {k: keep[k].cpu().numpy() for k in latched}
    fl = st["flags"].astype(np.uint32)
    coll, goal = (fl & nat.IN_COLLISION) != 0, (fl & nat.AT_GOAL) != 0
    rows = []
    for e, c in enumerate(cases):
        collision, all_at_goal = bool(coll[e].any()), bool(goal[e].all())
        rows.append({"num_agents": num_agents, "policy": policy, "test_case": c,
                     "total_reward": st["ep_reward"][e].copy(), "steps": int(finish[e]),
                     "time_to_goal": st["t"][e].copy(), "total_time_to_goal": float(st["t"][e].sum()),
                     "extra_time_to_goal": st["t"][e] - st["slt"][e], "collision": collision,
                     "all_at_goal": all_at_goal, "any_stuck": bool((~coll[e] & ~goal[e]).any()),
                     "outcome": "collision" if collision else "all_at_goal" if all_at_goal else "stuck"})
    return pd.DataFrame(rows)


def main():
    import pandas as pd
    frames = []
    for n in Config.NUM_AGENTS_TO_TEST:
        for pol in Config.POLICIES_TO_TEST:
            if pol not in policies:
                print("skipping %s: not available in this build" % pol)
                continue
            df = run_suite(pol, n, test_cases=range(Config.NUM_TEST_CASES))
            frames.append(df)
            print("%-16s %2d agents: %3d cases, %5.1f %% all at goal, %5.1f %% collision, mean extra time %.2f s" % (
                pol, n, len(df), 100 * df["all_at_goal"].mean(), 100 * df["collision"].mean(),
                np.mean([np.mean(x) for x in df["extra_time_to_goal"]])))
    return pd.concat(frames, ignore_index=True) if frames else None


if __name__ == "__main__":
    main()
    print("Experiment over.")
