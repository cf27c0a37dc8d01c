import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gym_collision_avoidance_amd import _native as nat, core
N, E = 10, 4096
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n10"]
sim = core.BatchedSim(core.make_params(E, N))
sim.set_plugins(nat.POL_RVO); sim.set_fixture_table(table); sim.reset_from_table()
L = nat.lib()
for _ in range(1800): sim.step()
buf = (C.c_ulonglong * (1024 * 16))()
names = ["top", "A1", "P1P2", "LP3coop", "post", "move", "publish", "P3", "A3", "P4", "A4", "copy", "LP2", "atan2", "wrap", "pre-atan2"]
for rep in range(3):
    sim.step()
    L.cagpu_debug_wgprof(buf)
    a = np.array(buf[:], dtype=np.float64).reshape(1024, 16)[:683, :16]
    tot = a.sum(1)
    order = np.argsort(tot)
    print("step", rep, "WG total cycles: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
    print("   mean per phase:", " ".join("%s=%.0f" % (n, v) for n, v in zip(names, a.mean(0))))
    print("   slowest 5 WGs :")
    for w in order[-5:]:
        print("     ", int(tot[w]), " ".join("%s=%.0f" % (n, v) for n, v in zip(names, a[w]) if v > 1500))
