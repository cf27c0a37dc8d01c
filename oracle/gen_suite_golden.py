#!/usr/bin/env python3
"""oracle/gen_suite_golden.py -- TEST INFRASTRUCTURE.  Runs the UNMODIFIED reference's own "benchmark" -- the full test
suite of experiments/src/run_full_test_suite.py:54-130, i.e. `run_episode` (experiments/src/env_utils.py:45-91) over the
500 fixture cases of an agent count, all agents RVO -- and records one row per case into tests/golden/suite_*.npz:

    outcome (0 collision / 1 all_at_goal / 2 stuck), steps, time_to_goal[N], extra_time_to_goal[N], total_reward[N],
    final flag word[N], final position[N, 2]

Suites: n10 (10_agents_500_cases.p), n4 (4_agents_500_cases.p) and `ragged4`: an env with MAX_NUM_AGENTS_IN_ENVIRONMENT = 4
whose episodes hold 2, 3 or 4 agents (case c = row c of the {2,3,4}[c % 3]-agent table) -- the reference's default reset
draws the agent count per episode (test_cases.py:224-227); the shipped `2_3_4_agents_500_cases.p` holds 3-agent cases only.

Like gen_golden.py this only works in the build container (needs /root/reference, the import stubs and the oracle's `rvo2`
module: the RVO stage is self-pinned, see oracle/orca_ref.h); the outputs are committed.  The Config object is an
import-time singleton, so every suite slice runs in its own subprocess (8 at a time).

Usage:  python oracle/gen_suite_golden.py [--suites n10 n4 ragged4] [--jobs 8]
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("CA_REFERENCE_ROOT", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")

SUITES = {  # name -> (config class, agent counts cycled over the cases)
    "n10": ("Bench10", (10,)),
    "n4": ("Swap4", (4,)),
    "ragged4": ("Swap4", (2, 3, 4)),
}
OUTCOME = {"collision": 0, "all_at_goal": 1, "stuck": 2}


def _flags(a):
    return (int(bool(a.is_at_goal)) | int(bool(a.was_at_goal_already)) << 1 | int(bool(a.in_collision)) << 2
            | int(bool(a.was_in_collision_already)) << 3 | int(bool(a.ran_out_of_time)) << 4 | int(bool(a.is_done)) << 5)


def worker(suite, lo, hi, out_path):
    cfg, counts = SUITES[suite]
    os.environ["GYM_CONFIG_PATH"] = os.path.join(HERE, "golden_configs.py")
    os.environ["GYM_CONFIG_CLASS"] = cfg
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path[:0] = [os.path.join(HERE, "stubs"), os.path.join(HERE, "_build"), REF]
    import warnings
    warnings.filterwarnings("ignore")
    import rvo2  # noqa: F401
    from gym_collision_avoidance.envs import Config
    from gym_collision_avoidance.envs import test_cases as tc
    from gym_collision_avoidance.envs.collision_avoidance_env import CollisionAvoidanceEnv
    from gym_collision_avoidance.experiments.src.env_utils import run_episode

    n_max = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT
    tables = {n: tc.preset_testCases(n, full_test_suite=True) for n in counts}
    env = CollisionAvoidanceEnv()
    rows = dict(case=[], num_agents=[], outcome=[], steps=[], time_to_goal=[], extra_time_to_goal=[], total_reward=[],
                flags=[], pos=[])
    pad = lambda v, fill=0.0: np.concatenate([np.asarray(v, np.float64), np.full(n_max - len(v), fill)])
    for c in range(lo, hi):
        n = counts[c % len(counts)]
        agents = tc.cadrl_test_case_to_agents(tables[n][c], policies="RVO", agents_dynamics="unicycle",
                                              agents_sensors=["other_agents_states"])
        env.set_agents(agents)
        env.reset()
        env.test_case_index = c
        stats, ags = run_episode(env)      # the reference's loop: while not terminated: env.step(None)
        assert stats["num_agents"] == n and len(ags) == n
        rows["case"].append(c)
        rows["num_agents"].append(n)
        rows["outcome"].append(OUTCOME[stats["outcome"]])
        rows["steps"].append(stats["steps"])
        rows["time_to_goal"].append(pad(stats["time_to_goal"]))
        rows["extra_time_to_goal"].append(pad(stats["extra_time_to_goal"]))
        rows["total_reward"].append(pad(np.asarray(stats["total_reward"], np.float64).reshape(-1)))
        # run_episode resets the env before it returns, so the agents' final flags / positions come from the history it
        # keeps: prev_episode_agents (collision_avoidance_env.py:262-268) is a deep copy taken at reset
        prev = env.prev_episode_agents
        rows["flags"].append(np.array([_flags(a) for a in prev] + [0] * (n_max - n), np.uint32))
        rows["pos"].append(np.array([a.pos_global_frame for a in prev] + [[0.0, 0.0]] * (n_max - n), np.float64))
    np.savez(out_path, **{k: np.array(v) for k, v in rows.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--suites", nargs="*", default=list(SUITES))
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--cases", type=int, default=500)
    ap.add_argument("--worker", nargs=4)
    a = ap.parse_args()
    if a.worker:
        worker(a.worker[0], int(a.worker[1]), int(a.worker[2]), a.worker[3])
        return
    subprocess.check_call(["make", "-C", HERE, "-s"])
    for suite in a.suites:
        with tempfile.TemporaryDirectory() as td:
            bounds = np.linspace(0, a.cases, a.jobs + 1).astype(int)
            procs = []
            for j in range(a.jobs):
                if bounds[j] == bounds[j + 1]:
                    continue
                path = os.path.join(td, "part%d.npz" % j)
                procs.append((path, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", suite,
                                                      str(bounds[j]), str(bounds[j + 1]), path])))
            parts = []
            for path, pr in procs:
                if pr.wait() != 0:
                    raise SystemExit("worker failed")
                with np.load(path) as z:
                    parts.append({k: z[k] for k in z.files})
        out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        assert np.array_equal(out["case"], np.arange(a.cases))
        out["config"] = np.array(SUITES[suite][0])
        os.makedirs(GOLD, exist_ok=True)
        np.savez_compressed(os.path.join(GOLD, "suite_%s.npz" % suite), **out)
        oc = np.bincount(out["outcome"], minlength=3) / float(a.cases)
        print("suite %s: %d cases, collision %.3f all_at_goal %.3f stuck %.3f, mean steps %.1f" %
              (suite, a.cases, oc[0], oc[1], oc[2], out["steps"].mean()))


if __name__ == "__main__":
    main()
