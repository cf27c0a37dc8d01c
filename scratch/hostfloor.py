import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gym_collision_avoidance_amd import _native as nat, core
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n10"]
for E in (16, 4096):
    sim = core.BatchedSim(core.make_params(E, 10))
    sim.set_plugins(nat.POL_RVO); sim.set_fixture_table(table); sim.reset_from_table()
    for _ in range(200): sim.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 2000
    for _ in range(n): sim.step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("E=%d: host enqueue %.1f us/step, total %.1f us/step" % (E, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    # raw ctypes call cost with cached args
    L = sim.lib; p, cs, co = C.byref(sim.p), C.byref(sim._cs), C.byref(sim._co)
    ar = C.byref(sim._ar); st = sim._stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): L.cagpu_step(p, cs, co, None, ar, st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("E=%d cached-args: host enqueue %.1f us/step, total %.1f us/step" % (E, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    t0 = time.perf_counter(); sim.rollout(n); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("E=%d rollout: total %.1f us/step" % (E, (t2 - t0) / n * 1e6))
