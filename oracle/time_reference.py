#!/usr/bin/env python3
"""oracle/time_reference.py -- TEST INFRASTRUCTURE.  Times the UNMODIFIED reference's own `env.step` (the
CPU baseline beside the GPU number) on the metric workload: 10-agent fixture cases, RVOPolicy (through the oracle's
`rvo2` module) + UnicycleDynamics + OtherAgentsStatesSensor, EvaluateConfig constants -- one `env.step(None)` per
step exactly like experiments/src/env_utils.py:45-52, a new fixture case whenever the episode ends
(run_full_test_suite.py:74-104 without the plotting).

  (i)  1 process  = 1 core (the reference is single-threaded);
  (ii) `nproc` independent processes, each stepping its own env, rates summed.

Only runs where /root/reference exists (the build container); the result is committed as
profiles/r02_reference_cpu.json and quoted by bench.py.

usage: python oracle/time_reference.py [--seconds 20] [--procs N]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("CA_REFERENCE_ROOT", "/root/reference")


def worker(args):
    rank, seconds = args
    os.environ["GYM_CONFIG_PATH"] = os.path.join(HERE, "golden_configs.py")
    os.environ["GYM_CONFIG_CLASS"] = "Bench10"
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path[:0] = [os.path.join(HERE, "stubs"), os.path.join(HERE, "_build"), REF]
    import warnings
    warnings.filterwarnings("ignore")
    import rvo2  # noqa: F401
    from gym_collision_avoidance.envs import test_cases as tc
    from gym_collision_avoidance.envs.collision_avoidance_env import CollisionAvoidanceEnv

    env = CollisionAvoidanceEnv()
    cases = tc.preset_testCases(10, full_test_suite=True)
    c = rank * 37 % 500

    def new_episode(c):
        agents = tc.cadrl_test_case_to_agents(cases[c], policies="RVO", agents_dynamics="unicycle",
                                              agents_sensors=["other_agents_states"])
        env.set_agents(agents)
        env.reset()
        return len(agents)

    n = new_episode(c)
    for _ in range(20):                      # warm-up
        env.step(None)
    agent_steps, episodes = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, over, _, _ = env.step(None)   # env_utils.py:50
        agent_steps += n
        if over:
            episodes += 1
            c = (c + 1) % 500
            n = new_episode(c)               # reset cost is part of the workload, as in the GPU auto-reset
    dt = time.perf_counter() - t0
    return agent_steps, dt, episodes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r02_reference_cpu.json"))
    a = ap.parse_args()
    assert os.path.isdir(REF), "the reference is not here (%s): this script only runs in the build container" % REF
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        s1, d1, e1 = pool.map(worker, [(0, a.seconds)])[0]
    with ctx.Pool(a.procs) as pool:
        res = pool.map(worker, [(r, a.seconds) for r in range(a.procs)])
    allv = sum(s / d for s, d, _ in res)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    out = {"what": "unmodified reference CollisionAvoidanceEnv.step(None), 10-agent fixture cases, RVOPolicy via the oracle's "
                   "rvo2 module, EvaluateConfig (DT=0.1), python %s / numpy" % sys.version.split()[0],
           "host": {"cpu": cpu, "logical_cores": os.cpu_count()},
           "one_process": {"agent_steps_per_s": s1 / d1, "env_steps_per_s": s1 / d1 / 10.0, "seconds": d1, "episodes": e1,
                           "cores": 1},
           "all_cores": {"agent_steps_per_s": allv, "processes": a.procs, "seconds": a.seconds,
                         "episodes": sum(e for _, _, e in res), "cores": a.procs}}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
