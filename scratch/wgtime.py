#!/usr/bin/env python3
"""scratch/wgtime.py -- per-workgroup wall-clock stamps of one launch of the PRODUCT-like step kernel (build variant
wgtime_fast: one start / end stamp per workgroup, no timers inside): start skew, duration distribution, and what the
slow workgroups have in common (linearProgram3 queue length, ORCA queries, an env that auto-resets).
    python gym_collision_avoidance_amd/build_native.py wgtime_fast; CAGPU_LIB=.../libcagpu_wgtime_fast.so python scratch/wgtime.py"""
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
buf = (C.c_ulonglong * (4096 * 8))()
rows = []
for rep in range(20):
    for _ in range(37):
        sim.step()
    lib.cagpu_debug_wgtime(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[:E // 4].astype(np.int64).copy()
    rows.append(a)
print(lib.cagpu_last_kernel().decode())
tick = 0.01  # us per tick (100 MHz)
dur_all, span_all = [], []
slow_n3, slow_reset, all_n3, all_reset, all_live, slow_live = [], [], [], [], [], []
for a in rows:
    t0, t1, t2, info = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    n3, live, rst = info & 0xFF, (info >> 8) & 0xFF, (info >> 16) & 1
    start = (t0 - t0.min()) * tick
    end = (t2 - t0.min()) * tick
    dur = (t2 - t0) * tick
    span_all.append(end.max())
    dur_all.append(dur)
    k = np.argsort(end)[-10:]
    slow_n3 += list(n3[k]); slow_reset += list(rst[k]); slow_live += list(live[k])
    all_n3 += list(n3); all_reset += list(rst); all_live += list(live)
    if len(span_all) == 1:
        print("one launch: start skew p50 %.2f p99 %.2f max %.2f us | duration mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f us | "
              "last end %.2f us | store tail (t2 - t1) mean %.2f us" % (
                  np.percentile(start, 50), np.percentile(start, 99), start.max(), dur.mean(), np.percentile(dur, 50),
                  np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), end.max(), ((t2 - t1) * tick).mean()))
        for lo, hi, name in ((0, 0, "n3 = 0"), (1, 1, "n3 = 1"), (2, 2, "n3 = 2"), (3, 4, "n3 = 3..4"), (5, 99, "n3 >= 5")):
            m = (n3 >= lo) & (n3 <= hi)
            if m.any():
                print("   %-10s %4d workgroups: duration mean %.2f p90 %.2f max %.2f" % (name, m.sum(), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max()))
        lp3 = a[:, 6] * tick
        for lo, hi, name in ((0, 0, "n3 = 0"), (1, 1, "n3 = 1"), (2, 2, "n3 = 2"), (3, 4, "n3 = 3..4"), (5, 99, "n3 >= 5")):
            m = (n3 >= lo) & (n3 <= hi)
            if m.any():
                print("   %-10s linearProgram3 pass: mean %.2f us p50 %.2f p90 %.2f max %.2f" % (name, lp3[m].mean(), np.percentile(lp3[m], 50), np.percentile(lp3[m], 90), lp3[m].max()))
        for v in (0, 1):
            m = rst == v
            if m.any():
                print("   reset=%d    %4d workgroups: duration mean %.2f p90 %.2f max %.2f" % (v, m.sum(), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max()))
        for lo, hi in ((0, 16), (17, 24), (25, 28), (29, 40)):
            m = (live >= lo) & (live <= hi)
            if m.any():
                print("   queries %2d..%2d %4d workgroups: duration mean %.2f p90 %.2f max %.2f" % (lo, hi, m.sum(), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max()))
d = np.concatenate(dur_all)
print("20 launches: first start -> last end: mean %.2f us (min %.2f max %.2f); workgroup duration mean %.2f p50 %.2f p90 %.2f p99 %.2f p99.9 %.2f" % (
    np.mean(span_all), np.min(span_all), np.max(span_all), d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), np.percentile(d, 99.9)))
print("the 10 workgroups that end last in each launch: n3 mean %.2f (all: %.2f), with a reset %.2f (all: %.2f), queries %.1f (all: %.1f)" % (
    np.mean(slow_n3), np.mean(all_n3), np.mean(slow_reset), np.mean(all_reset), np.mean(slow_live), np.mean(all_live)))

# ---- phase boundaries of the last launch (10 ns stamps on thread 0 of every workgroup; slots as in prof_phases.py)
SLOTS = {0: "entry / loads issued", 1: "A1 bodies + barrier", 2: "P2 rank, half-planes + barrier", 12: "P2b 1-D programmes + barrier",
         15: "scan (wave 0) + queue + barrier", 3: "linearProgram3 pass", 13: "-", 14: "-", 4: "A2c: policy post (atan2, sqrt, clip)",
         5: "A2c: move (sincos) + bookkeeping", 6: "publish + ego frame + barrier", 7: "P3 pair dist / keys + barrier",
         8: "P4 round 1 + A3 (wave 0)", 9: "A4 prologue", 10: "A4 + barrier(or)", 11: "reset copy / end"}
ph = (C.c_uint * (4096 * 16))()
lib.cagpu_debug_wgphase(ph)
P = np.frombuffer(ph, dtype=np.uint32).reshape(4096, 16)[:E // 4].astype(np.float64) * tick
a = rows[-1]
n3 = a[:, 3] & 0xFF
light = n3 == 0
print("phase (us)                                  all: mean  p90 |  n3 = 0: mean | n3 > 0: mean | slowest 1 %: mean")
slow = np.argsort(P.sum(axis=1))[-max(len(P) // 100, 1):]
for sl in (0, 1, 2, 12, 15, 3, 4, 5, 6, 7, 8, 9, 10, 11):
    c = P[:, sl]
    print("  %-40s %5.2f %5.2f | %5.2f | %5.2f | %5.2f" % (SLOTS[sl], c.mean(), np.percentile(c, 90), c[light].mean(), c[~light].mean(), c[slow].mean()))
print("  %-40s %5.2f" % ("sum", P.sum(axis=1).mean()))

it3 = (a[:, 3] >> 24) & 0xFF   # outer iterations of the queue entries wave 0 solved (entry 0, 4, ...)
for nit in (1, 2, 3, 4):
    m = (it3 == nit) & (n3 >= 1) & (n3 <= 4)
    if m.any():
        print("  wave 0's linearProgram3 entry with %d outer iteration(s): %4d workgroups: setup %.2f us, lp3_wave8 %.2f us (p90 %.2f)" % (
            nit, m.sum(), P[m, 13].mean(), P[m, 14].mean(), np.percentile(P[m, 14], 90)))

cyc = a[:, 7].astype(np.float64); life = (a[:, 2] - a[:, 0]).astype(np.float64) * tick
print("shader clock during the launch: %.0f MHz (clock64 cycles / wall-clock us over each workgroup's life, median)" % np.median(cyc / life))
