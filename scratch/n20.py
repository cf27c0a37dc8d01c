#!/usr/bin/env python3
"""scratch/n20.py -- step time at 4096 envs x N RVO agents (N from env, default 20); with the knobs build CAGPU_NO_NC=1 selects
the generic instantiation instead of the compile-time-N one"""
import numpy as np, torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_collision_avoidance_amd import _native as nat, core
N = int(os.environ.get("N", "20"))
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n%d" % N]
sim = core.BatchedSim(core.make_params(4096, N))
sim.set_plugins(nat.POL_RVO); sim.set_fixture_table(table); sim.reset_from_table()
for _ in range(300): sim.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): sim.step()
e1.record(); torch.cuda.synchronize()
print("RVO 4096 x %d: %.1f us / step" % (N, e0.elapsed_time(e1) * 1e3 / 200), nat.lib().cagpu_last_kernel().decode())
