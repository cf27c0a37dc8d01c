import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gym_collision_avoidance_amd import _native as nat, core
E, N = 4096, 20
sim = core.BatchedSim(core.make_params(E, N, max_obs=19, sort_mode=1))
sim.set_plugins(nat.POL_GA3C_CADRL); sim.load_ga3c()
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n20"]
sim.reset(table[np.arange(E) % 500])
def t(n=50):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): sim.ga3c()
    e0.record()
    for _ in range(n): sim.ga3c()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
full = t()
for k in (0, 1, 5, 10, 19):
    sim.obs[..., 1] = float(k)
    print("seq_len", k, "%.1f us" % t())
print("full (19 everywhere)", "%.1f us" % full)
macs_fc = 68 * 256 + 2 * 256 * 256 + 256 * 11
macs_l = 71 * 256
B = E * N
print("FC-only flops %.1f GF, per LSTM step %.1f GF" % (2 * macs_fc * B / 1e9, 2 * macs_l * B / 1e9))
