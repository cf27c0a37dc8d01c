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
for _ in range(steps): sim.step()
L.cagpu_debug_prof(buf, 1)
print("LP3 agents / RVO queries:", buf[14], buf[15], buf[14]/max(1,buf[15])); print("slot12 (lp2 part):", buf[12]/((E+5)//6)/steps)
names = ["0 loop top", "1 A1 (bodies, or-barrier)", "2 P1+P2 (dist, rank, half-planes)", "3 A2a pref + LP", "4 A2b post (atan2, wrap, sqrt)",
         "5 A2c move (sincos)", "6 publish + ego + barrier", "7 P3 pair dist/keys + barrier", "8 A3 reward", "9 P4 rank/emit + barrier",
         "10 A4 done/reset + or-barrier", "11 copy-out + barrier"]
wg = (E + 5) // 6
tot = 0
for i, n in enumerate(names):
    c = buf[i] / wg / steps
    tot += c
    print("%-45s %9.0f cycles/step/WG" % (n, c))
print("total", tot, "cycles/step =", tot / 2.1e3, "us @2.1GHz")
