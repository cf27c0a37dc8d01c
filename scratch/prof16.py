import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gym_collision_avoidance_amd import _native as nat, core
N, E = 10, 4096
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n10"]
sim = core.BatchedSim(core.make_params(E, N))
sim.set_plugins(nat.POL_RVO); sim.set_fixture_table(table); sim.reset_from_table()
L = nat.lib()
buf = (C.c_ulonglong * 16)()
for _ in range(1500): sim.step()
L.cagpu_debug_prof(buf, 1)
steps = 500
import time
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(steps): sim.step()
torch.cuda.synchronize(); print('wall us/step', (time.perf_counter()-t0)/steps*1e6)
L.cagpu_debug_prof(buf, 1)
names = ["0 loop top", "1 S1 bodies+pref + barrier", "2 S2 dist/rank/half-plane (wave 0 of WG)", "3 S3 group LP + barrier", "4 S4 post/move (atan2,sincos)",
         "5 ego frames + barrier", "6 S5 pair gaps/keys/ranks/rows + barrier", "7 S6 reward + barrier", "8 S7 game over/reset + barriers", "9 copy-out + barrier"]
epw = int(os.environ.get("CAGPU_EPW", "2"))
wg = (E + epw - 1) // epw
tot = 0
for i, n in enumerate(names):
    c = buf[i] / wg / steps; tot += c
    print("%-45s %9.0f cycles/step/WG" % (n, c))
print("total %.0f cycles/step/WG = %.1f us @2.4GHz" % (tot, tot / 2.4e3))
