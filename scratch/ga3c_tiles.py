#!/usr/bin/env python3
"""scratch/ga3c_tiles.py -- time of cagpu_ga3c for a forced tile height (knobs build: CAGPU_GA3C_TILE=64|48|32|0) with
every agent alive (just after a reset) and at the workload's steady state: calibrates tile_rows() in cagpu_ga3c.inc."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_collision_avoidance_amd import _native as nat, core
E, N = 4096, 20
table = np.load(os.path.join(os.path.dirname(nat.HERE), "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n20"]
sim = core.BatchedSim(core.make_params(E, N, max_obs=19, sort_mode=nat.SORT_CLOSEST_LAST))
sim.set_plugins(nat.POL_GA3C_CADRL, nat.DYN_UNICYCLE)
sim.load_ga3c()
sim.set_fixture_table(table)
sim.reset_from_table()
def t_ga3c(n=30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): sim.ga3c()
    e0.record()
    for _ in range(n): sim.ga3c()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
a = t_ga3c(); ra = sim.ga3c_rows()
for _ in range(int(os.environ.get("STEPS", "300"))): sim.step()
b = t_ga3c(); rb = sim.ga3c_rows()
print("tile %s: all alive %d rows %.1f us | steady state %d rows %.1f us" % (os.environ.get("CAGPU_GA3C_TILE", "auto"), ra, a, rb, b))
