import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gym_collision_avoidance_amd import _native as nat, core
E, N = 4096, 20
sim = core.BatchedSim(core.make_params(E, N, max_obs=19, sort_mode=1))
sim.set_plugins(nat.POL_GA3C_CADRL); sim.load_ga3c()
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n20"]
sim.reset(table[np.arange(E) % 500])
L = nat.lib(); buf = (C.c_ulonglong * 16)()
for _ in range(5): sim.ga3c()
L.cagpu_debug_prof(buf, 1)
n = 20
for _ in range(n): sim.ga3c()
L.cagpu_debug_prof(buf, 1)
wgs = (E * N + 63) // 64
names = ["0 flags/zero", "1 weights + obs load/normalise", "2 LSTM (19 steps)", "3 concat + layer1", "4 layer2", "5 fc1", "6 logits"]
tot = 0
for i, nm in enumerate(names):
    c = buf[i] / wgs / n; tot += c
    print("%-34s %9.0f cycles/WG" % (nm, c))
print("total %.0f cycles/WG; MFMA-only would be %d" % (tot, 7872 * 32))
