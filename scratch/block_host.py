"""scratch/block_host.py -- where the host's share of a 20-step timed block goes (median over blocks, us)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_collision_avoidance_amd import core
dev = torch.device("cuda", 0)
sim, table, N, K = bench.build_workload("rvo10", 4096, dev)
sim.enable_lookahead(20, fresh=True)
stamps = {}
orig_fill = core.BatchedSim._la_fill
lib = sim.lib
orig_ring = lib.cagpu_rollout_ring
class Wrap:
    def __call__(self, *a):
        stamps["call_in"] = time.perf_counter()
        r = orig_ring(*a)
        stamps["call_out"] = time.perf_counter()
        return r
sim.lib = type("L", (), {})()
for n in dir(lib):
    if n.startswith("cagpu_"):
        setattr(sim.lib, n, getattr(lib, n))
sim.lib.cagpu_rollout_ring = Wrap()
for _ in range(400):
    sim.step_lookahead()
torch.cuda.synchronize()
rows = []
for b in range(600):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step_lookahead()
    t1 = time.perf_counter()
    for _ in range(19):
        sim.step_lookahead()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((stamps["call_in"] - t0, stamps["call_out"] - stamps["call_in"], t1 - stamps["call_out"], t2 - t1, t3 - t2, t3 - t0))
a = np.median(np.array(rows), axis=0) * 1e6
print("before the C call %.1f | C call (checks + hipLaunch) %.1f | rest of the fill (probe, next ring prepared) %.1f | 19 more hand-outs %.1f | synchronize %.1f | block %.1f us" % tuple(a))
