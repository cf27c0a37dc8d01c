#!/usr/bin/env python3
"""scratch/pipetime_multi.py -- stage stamps (pipetime_fast build) of the LAST step of an n-step launch: what does the chain of the
workgroups a 20-step launch waits for look like when they run nearly alone at its end?
    CAGPU_LIB=gym_collision_avoidance_amd/libcagpu_pipetime_fast.so python scratch/pipetime_multi.py [L]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gym_collision_avoidance_amd import _native as nat  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sim, table, N, K = bench.build_workload("rvo10", 4096, torch.device("cuda", 0))
lib = nat.lib()
sim.rollout(300)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (1024 * 32))()
W0 = ["start", "loaded", "B_A", "move", "published", "B_P", "P3 arrived", "A3 done", "(A3)", "A4 done", "A4 signalled", "(same)", "stored"]
W4 = ["B_P", "P2 done", "P2 all", "P2b done", "P2b all", "scan done", "queue seen", "lp3 done", "lp3 all + A4", "post done"]
tick = 0.01
groups = {"last 20 finishers": [], "finishers 400..600": []}
info = {"last 20 finishers": [], "finishers 400..600": []}
for rep in range(12):
    sim.rollout(L)
    lib.cagpu_debug_pipetime(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 32).astype(np.int64).copy()
    if rep < 2:
        continue
    end = np.maximum(a[:, 12], a[:, 25])
    order = np.argsort(end)
    for name, sel in (("last 20 finishers", order[-20:]), ("finishers 400..600", order[400:600])):
        w0 = a[sel][:, 3:12]          # move .. (same): the steady-state part of the last step
        w4 = a[sel][:, 16:26].copy()
        noq = (a[sel][:, 13] & 0xFF) == 0
        w4[noq, 7] = w4[noq, 6]
        s0, s4 = np.diff(w0, axis=1) * tick, np.diff(w4, axis=1) * tick
        groups[name].append((s0, s4, (a[sel][:, 25] - a[sel][:, 16]) * tick, (a[sel][:, 11] - a[sel][:, 3]) * tick))
        info[name].append(np.stack([a[sel][:, 13] & 0xFF, (a[sel][:, 13] >> 8) & 0xFF, (a[sel][:, 13] >> 24) & 0xFF, (a[sel][:, 13] >> 32) & 0xFF], axis=1))
print(lib.cagpu_last_kernel().decode(), "L =", L)
for name in groups:
    s0 = np.concatenate([g[0] for g in groups[name]]); s4 = np.concatenate([g[1] for g in groups[name]])
    o_half = np.concatenate([g[2] for g in groups[name]]); w0t = np.concatenate([g[3] for g in groups[name]])
    inf = np.concatenate(info[name])
    print("== %s: O half (B_P -> post done) %.2f us, wave 0 (move -> A4 signalled) %.2f us; lp3 queue %.2f, planned %.1f, acted lines sum %.2f max %.2f" % (
        name, o_half.mean(), w0t.mean(), inf[:, 0].mean(), inf[:, 1].mean(), inf[:, 2].mean(), inf[:, 3].mean()))
    print("   wave 0: " + "  ".join("%s %.2f" % (W0[3 + i + 1], s0[:, i].mean()) for i in range(s0.shape[1])))
    print("   wave 4: " + "  ".join("%s %.2f" % (W4[i + 1], s4[:, i].mean()) for i in range(s4.shape[1])))
    for lo, hi in ((0, 0), (1, 1), (2, 3), (4, 9)):
        m = (inf[:, 0] >= lo) & (inf[:, 0] <= hi)
        if m.any():
            print("   lp3 queue %d..%d: %4.0f %%  O half %.2f us  (lp3 phase %.2f)" % (lo, hi, 100 * m.mean(), o_half[m].mean(), (s4[m, 6] + s4[m, 7]).mean()))
