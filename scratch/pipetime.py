#!/usr/bin/env python3
"""scratch/pipetime.py -- stage-boundary wall-clock stamps of the pipelined step kernel (build variant pipetime_fast):
where a workgroup's step goes, how the slowest workgroups of a launch differ from the mean.
    python gym_collision_avoidance_amd/build_native.py pipetime_fast
    CAGPU_LIB=gym_collision_avoidance_amd/libcagpu_pipetime_fast.so python scratch/pipetime.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_collision_avoidance_amd import _native as nat, core  # noqa: E402

E = int(os.environ.get("E", "4096"))
table = np.load(os.path.join(os.path.dirname(nat.HERE), "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n10"]
sim = core.BatchedSim(core.make_params(E, 10))
sim.set_plugins(nat.POL_RVO)
sim.set_fixture_table(table)
sim.reset_from_table()
lib = nat.lib()
for _ in range(400):
    sim.step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (1024 * 32))()
rows = []
for rep in range(20):
    for _ in range(37):
        sim.step()
    lib.cagpu_debug_pipetime(buf)
    rows.append(np.frombuffer(buf, dtype=np.uint64).reshape(1024, 32)[:E // 4].astype(np.int64).copy())
print(lib.cagpu_last_kernel().decode())
tick = 0.01
W0 = ["start", "loaded", "B_A", "moved(pre-move pt)", "published", "B_P", "P3 arrived", "A3 done", "(A3 done)", "A4 done", "A4 signalled", "(same)", "stored"]
W4 = ["B_P", "P2 done", "P2 of all waves", "P2b done", "P2b of all waves", "scan done", "queue seen", "lp3 done", "lp3 of all + A4 seen", "post done"]
dur, span, seg0, seg4, slow0, slow4, n3s, lives = [], [], [], [], [], [], [], []
skew, lastdur, laststart = [], [], []
acts, actmax, slow_info = [], [], []
for a in rows:
    t0 = a[:, 0]
    end = np.maximum(a[:, 12], a[:, 25])
    d = (end - t0) * tick
    dur.append(d)
    span.append((end.max() - t0.min()) * tick)
    skew.append((np.percentile(t0, [50, 90, 99, 100]) - t0.min()) * tick)
    kk = np.argsort(end)[-5:]
    lastdur.append(d[kk]); laststart.append((t0[kk] - t0.min()) * tick)
    w0 = a[:, 0:13].copy()
    s0 = np.diff(w0, axis=1) * tick
    w4 = a[:, 16:26].copy()
    noq = (a[:, 13] & 0xFF) == 0
    w4[noq, 7] = w4[noq, 6]
    s4 = np.diff(w4, axis=1) * tick
    seg0.append(s0); seg4.append(s4)
    k = np.argsort(end)[-10:]
    slow0.append(s0[k]); slow4.append(s4[k])
    n3s.append(a[:, 13] & 0xFF); lives.append((a[:, 13] >> 8) & 0xFF)
    acts.append((a[:, 13] >> 24) & 0xFF); actmax.append((a[:, 13] >> 32) & 0xFF)
    slow_info.append(np.stack([a[k, 13] & 0xFF, (a[k, 13] >> 8) & 0xFF, (a[k, 13] >> 24) & 0xFF, (a[k, 13] >> 32) & 0xFF, (a[k, 13] >> 16) & 0xFF], axis=1))
dur = np.concatenate(dur); seg0 = np.concatenate(seg0); seg4 = np.concatenate(seg4)
slow0 = np.concatenate(slow0); slow4 = np.concatenate(slow4); n3s = np.concatenate(n3s); lives = np.concatenate(lives)
print("workgroup duration (start -> last store): mean %.2f p50 %.2f p90 %.2f p99 %.2f p99.9 %.2f max %.2f us; first start -> last end per launch: mean %.2f" % (
    dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), np.percentile(dur, 99.9), dur.max(), np.mean(span)))
print("start skew after the first workgroup: p50 %.2f p90 %.2f p99 %.2f max %.2f us; the 5 last-finishing workgroups of a launch: started at +%.2f, ran %.2f us (mean)" % (tuple(np.mean(skew, axis=0)) + (np.mean(laststart), np.mean(lastdur))))
print("wave 0 segments (mean all | mean of the 10 last-finishing workgroups per launch):")
for i in range(12):
    print("   %-22s -> %-22s %6.2f | %6.2f" % (W0[i], W0[i + 1], seg0[:, i].mean(), slow0[:, i].mean()))
print("wave 4 segments:")
for i in range(9):
    print("   %-22s -> %-22s %6.2f | %6.2f" % (W4[i], W4[i + 1], seg4[:, i].mean(), slow4[:, i].mean()))
for lo, hi in ((0, 0), (1, 1), (2, 2), (3, 4), (5, 99)):
    m = (n3s >= lo) & (n3s <= hi)
    if m.any():
        print("   lp3 queue %d..%d: %5.1f %% of the workgroups, duration mean %.2f p90 %.2f" % (lo, hi, 100 * m.mean(), dur[m].mean(), np.percentile(dur[m], 90)))
actmax = np.concatenate(actmax); slow_info = np.concatenate(slow_info)
for v in range(0, 5):
    m = actmax == v
    if m.any():
        print("   most lines linearProgram3 acted on for one agent = %d: %5.1f %%, duration mean %.2f p90 %.2f" % (v, 100 * m.mean(), dur[m].mean(), np.percentile(dur[m], 90)))
print("the 10 last-finishing workgroups per launch: queue length mean %.2f, planned agents mean %.1f (> 28: %.0f %%), acted lines sum %.2f max-per-agent hist %s, resets %.0f %%" % (slow_info[:, 0].mean(), slow_info[:, 1].mean(), 100 * (slow_info[:, 1] > 28).mean(), slow_info[:, 2].mean(), np.round(np.bincount(slow_info[:, 3], minlength=5) / len(slow_info), 2), 100 * (slow_info[:, 4] > 0).mean()))
for lo, hi in ((0, 16), (17, 24), (25, 28), (29, 40)):
    m = (lives >= lo) & (lives <= hi)
    if m.any():
        print("   planned agents %d..%d: %5.1f %%, duration mean %.2f p90 %.2f" % (lo, hi, 100 * m.mean(), dur[m].mean(), np.percentile(dur[m], 90)))
