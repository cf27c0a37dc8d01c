#!/usr/bin/env python3
"""scratch/envapi_probe.py -- host wall clock per CollisionAvoidanceEnv(num_envs=4096).step(None) for a list of `lookahead`
values (whole rings timed, device idle at both ends), and the same rings through core.BatchedSim.step_lookahead directly."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GYM_CONFIG_CLASS", "EvaluateConfig")
import bench
from gym_collision_avoidance_amd.envs import Config
from gym_collision_avoidance_amd.envs.collision_avoidance_env import CollisionAvoidanceEnv
dev = torch.device("cuda", 0)
E, N = 4096, 10
Config.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
Config.MAX_NUM_OTHER_AGENTS_OBSERVED = N - 1
for la in [int(x) for x in (sys.argv[1:] or ["32", "64", "93", "0"])]:
    env = CollisionAvoidanceEnv(num_envs=E, device=str(dev), lookahead=la)
    env.set_fixture_suite(N)
    env.reset()
    L = max(la, 1)
    for _ in range(8 * L if la else 300):     # (the adaptive ring has reached its length: 8, 16, ... la)
        env.step(None)
    ring = env._sim._la
    while ring is not None and ring["t"] < ring["len"]:   # stand at a ring boundary
        env.step(None)
    torch.cuda.synchronize()
    n = (640 // L) * L if la else 640
    t0 = time.perf_counter()
    for _ in range(n):
        env.step(None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("env API  lookahead %3d: %6.2f us per step (host loop alone %5.2f), ring now %s" % (la, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6,
                                                                                               None if ring is None else ring["len"]))
    del env
    if la:
        sim, table, _, _ = bench.build_workload("rvo10", E, dev)
        sim.enable_lookahead(la, fresh=True)
        for _ in range(4 * la):
            sim.step_lookahead()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            sim.step_lookahead()
        torch.cuda.synchronize()
        print("BatchedSim ring   %3d: %6.2f us per step" % (la, (time.perf_counter() - t0) / n * 1e6))
