#!/usr/bin/env python3
"""scratch/prof_phases.py -- in-kernel phase timers of the step kernel (needs an -DCAGPU_ABLATE build:
python gym_collision_avoidance_amd/build_native.py ablate_fast; CAGPU_LIB=.../libcagpu_ablate_fast.so).
Prints mean cycles per phase per workgroup per step and the per-workgroup distribution of the last launch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_collision_avoidance_amd import _native as nat, core  # noqa: E402

SLOTS = {0: "step entry / loop", 1: "A1 bodies + barrier", 2: "P2 rank, half-planes (+ pref on wave 0)", 12: "P2b 1-D programmes of all lines + barrier",
         3: "LP3 pass", 15: "LP2 scan (wave 0) + queue + barrier", 13: "A2c: atan2", 14: "A2c: wrap", 4: "A2c: rest of policy post",
         5: "A2c: move (sincos) + bookkeeping", 6: "publish + ego frame + barrier", 7: "P3 pair dist / keys + barrier",
         8: "P4 round 1 + A3 reward (wave 0)", 9: "wait for P4 round 2 + barrier", 10: "A4 done / reset + barrier(or)", 11: "reset-obs copy / copy-out / end sync"}
ORDER = [0, 1, 2, 12, 15, 3, 13, 14, 4, 5, 6, 7, 8, 9, 10, 11]

E = int(os.environ.get("E", "4096"))
steps = int(os.environ.get("STEPS", "200"))
mode = os.environ.get("MODE", "step")
NA = int(os.environ.get("N", "10"))
table = np.load(os.path.join(os.path.dirname(nat.HERE), "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n%d" % NA]
sim = core.BatchedSim(core.make_params(E, NA))
sim.set_plugins(getattr(nat, os.environ.get("POLICY", "POL_RVO")))
sim.set_fixture_table(table)
sim.reset_from_table()
lib = nat.lib()
for _ in range(150):
    sim.step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib.cagpu_debug_prof(buf, 1)
if mode == "rollout":
    sim.rollout(steps)
else:
    for _ in range(steps):
        sim.step()
torch.cuda.synchronize()
lib.cagpu_debug_prof(buf, 0)
print(lib.cagpu_last_kernel().decode())
wgs = int(lib.cagpu_last_kernel().decode().split("grid=")[1].split()[0])
tot = 0.0
for sl in ORDER:
    v = buf[sl] / (wgs * steps)
    tot += v
    print("%-45s %9.0f" % (SLOTS[sl], v))
print("%-45s %9.0f cycles = %.2f us at 2.4 GHz" % ("sum (mean per workgroup per step)", tot, tot / 2400.0))
wg = (C.c_ulonglong * (1024 * 16))()
lib.cagpu_debug_wgprof(wg)
a = np.frombuffer(wg, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
if mode == "rollout":
    a /= steps
n = min(wgs, 1024)
t = a[:n].sum(axis=1)
print("per-workgroup total (last launch): mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    t.mean(), np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max()))
for sl in ORDER:
    c = a[:n, sl]
    print("  %-40s mean %5.0f p50 %5.0f p90 %5.0f p99 %5.0f max %5.0f" % (SLOTS[sl][:40], c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max()))
# which phases make the slowest workgroups slow: mean of each phase over the slowest 1 % minus the overall mean
slow = np.argsort(t)[-max(n // 100, 1):]
print("slowest 1 %% of the workgroups (total %.0f vs mean %.0f): excess per phase" % (t[slow].mean(), t.mean()))
for sl in ORDER:
    print("  %-40s %+6.0f" % (SLOTS[sl][:40], a[slow, sl].mean() - a[:n, sl].mean()))
