#!/usr/bin/env python3
"""scratch/big_rate.py -- throughput of the large-env kernel (csrc/cagpu_big.inc): E envs x N > 64 RVO agents, make_testcase_huge
scenes, auto-reset; agent-steps/s from HIP events."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GYM_CONFIG_CLASS", "EvaluateConfig")
from gym_collision_avoidance_amd import _native as nat, core
from gym_collision_avoidance_amd.envs import test_cases as tc
out = []
for N, E in ((100, 512), (100, 64), (128, 512), (256, 128)):
    np.random.seed(N)
    table = tc.make_testcase_huge(8, N, side_length=25 if N <= 100 else 2.5 * np.sqrt(N) + 3)
    g = core.BatchedSim(core.make_params(E, N, max_obs=19))
    g.set_plugins(nat.POL_RVO)
    g.set_fixture_table(table)
    g.reset_from_table()
    for _ in range(30):
        g.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        g.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    out.append({"agents": N, "envs": E, "ms_per_step": ms, "agent_steps_per_s": E * N / (ms * 1e-3),
                "kernel": nat.lib().cagpu_last_kernel().decode(), "workspace_MB": g._workspace.numel() / 1e6})
print(json.dumps(out))
