#!/usr/bin/env python3
"""scratch/steptime.py -- per-step wall-clock stamps of every workgroup of the n-step kernel (build variant steptime_fast):
are the first steps of a launch slow for everybody (a ramp), or are some workgroups persistently slow (what a short launch
waits for at its end)?  And what do the slow ones hold?
    python -m gym_collision_avoidance_amd.build_native steptime_fast
    CAGPU_LIB=gym_collision_avoidance_amd/libcagpu_steptime_fast.so python scratch/steptime.py [L]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gym_collision_avoidance_amd import _native as nat  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E = 4096
sim, table, N, K = bench.build_workload("rvo10", E, torch.device("cuda", 0))
lib = nat.lib()
sim.rollout(300)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (1024 * 66))()
tick = 0.01   # us
durs, spans, infos, cus = [], [], [], []
BAL = os.environ.get("BALANCE", "")
T4 = E // 4


def balance():
    """envs physically re-ordered so that every tile holds about the same number of planned agents (scratch/balance_probe.py)"""
    f = sim.state["flags"]
    cnt = ((f & (nat.AT_GOAL | nat.OUT_OF_TIME | nat.IN_COLLISION | nat.ABSENT)) == 0).sum(dim=1)
    srt = torch.argsort(cnt, descending=True, stable=True)
    r = torch.arange(E, device=f.device)
    tile, j = r // 4, r % 4
    rank = torch.where(j % 2 == 0, j * T4 + tile, (j + 1) * T4 - 1 - tile)
    order = srt[rank] if BAL == "1" else srt
    for n, t in sim._state.items():
        t.copy_(t[order])
    torch.cuda.synchronize()


tile_map = torch.arange(T4, dtype=torch.int32, device=torch.device("cuda", 0))


def deal_tiles():
    """BALANCE=cu (round 6; needs the `tilemap` plumbing of commit "tile-map probe" -- KArgs.tile_map + cagpu_debug_set_tile_map,
    removed from the product sources after the measurement, profiles/r06_kernel_geometry.md): tiles stay as they are, but the block -> tile table deals them to the CUs (block b runs on
    CU b mod 256) snake-wise by planned-agent count, so that every CU's four tiles carry about the same total"""
    f = sim.state["flags"]
    cnt = ((f & (nat.AT_GOAL | nat.OUT_OF_TIME | nat.IN_COLLISION | nat.ABSENT)) == 0).sum(dim=1).view(T4, 4).sum(dim=1)
    order = torch.argsort(cnt, descending=True, stable=True)          # order[r] = the tile of rank r
    r = torch.arange(T4, device=f.device)
    rnd, pos = r // 256, r % 256
    block = torch.where(rnd % 2 == 0, pos, 255 - pos) + 256 * rnd
    tile_map[block] = order.to(torch.int32)
    torch.cuda.synchronize()
    lib.cagpu_debug_set_tile_map(C.c_void_p(tile_map.data_ptr()))


for rep in range(12):
    if BAL == "cu":
        deal_tiles()
    elif BAL:
        balance()
    sim.rollout(L)
    lib.cagpu_debug_steptime(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 66).copy()
    t = (a[:, :L + 1] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
    nlive = ((a[:, 1:L + 1] >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64)
    n3 = ((a[:, 1:L + 1] >> np.uint64(56)) & np.uint64(0xFF)).astype(np.int64)
    if rep < 2:
        continue
    durs.append(np.diff(t, axis=1) * tick)          # [1024, L] us per step (step 0 includes the load prologue)
    spans.append(((t[:, L].max() - t[:, 0].min()) * tick, (t[:, L] - t[:, 0]).mean() * tick, (t[:, 0].max() - t[:, 0].min()) * tick))
    infos.append((nlive, n3))
    cus.append(a[:, 64].astype(np.int64))
print(lib.cagpu_last_kernel().decode())
d = np.stack(durs)                                   # [reps, 1024, L]
sp = np.array(spans)
print("L = %d: launch span (first start -> last end) %.1f us = %.3f us / step; mean workgroup lifetime %.1f us; start skew %.2f us" % (
    L, sp[:, 0].mean(), sp[:, 0].mean() / L, sp[:, 1].mean(), sp[:, 2].mean()))
print("mean step duration by position in the launch (us):")
print("  " + " ".join("%5.2f" % x for x in d.mean(axis=(0, 1))))
print("p99 over workgroups by position:")
print("  " + " ".join("%5.2f" % x for x in np.percentile(d, 99, axis=1).mean(axis=0)))
tot = d[:, :, 1:].sum(axis=2)                        # per workgroup, steps 1 .. L-1
print("per-workgroup total of steps 1..%d: mean %.1f sd %.2f max %.1f (max - mean = %.1f us = what the launch waits for)" % (
    L - 1, tot.mean(), tot.std(axis=1).mean(), tot.max(axis=1).mean(), (tot.max(axis=1) - tot.mean(axis=1)).mean()))
h = (L - 1) // 2
first, second = d[:, :, 1:1 + h].mean(axis=2), d[:, :, 1 + h:1 + 2 * h].mean(axis=2)
print("persistence: correlation of a workgroup's mean pace in the first and the second half of a launch: %.3f" % np.mean(
    [np.corrcoef(first[r], second[r])[0, 1] for r in range(d.shape[0])]))
pace = d[:, :, 1:].mean(axis=2)                      # [reps, 1024]
nl = np.stack([i[0][:, 1:].mean(axis=1) for i in infos])
q3 = np.stack([i[1][:, 1:].mean(axis=1) for i in infos])
print("correlation of a workgroup's pace with its planned agents %.3f, with its linearProgram3 queue %.3f" % (
    np.mean([np.corrcoef(pace[r], nl[r])[0, 1] for r in range(len(pace))]), np.mean([np.corrcoef(pace[r], q3[r])[0, 1] for r in range(len(pace))])))
# per-step effect: a step's duration against what the tile held in that step
dd, nn, qq = d[:, :, 1:].reshape(-1), np.stack([i[0][:, 1:] for i in infos]).reshape(-1), np.stack([i[1][:, 1:] for i in infos]).reshape(-1)
for lo, hi in ((0, 12), (12, 20), (20, 28), (28, 34), (34, 41)):
    m = (nn >= lo) & (nn < hi)
    print("  steps of tiles with %2d..%2d planned agents: %7d steps, mean %.2f us" % (lo, hi - 1, m.sum(), dd[m].mean() if m.any() else float("nan")))
for q in range(0, 8):
    m = qq == q
    if m.sum() > 50:
        print("  steps with %d linearProgram3 agents: %7d steps, mean %.2f us" % (q, m.sum(), dd[m].mean()))
# per CU: the four workgroups that share it
cu = ((cus[0] >> 16) << 16) | (cus[0] & 0xFF00)      # xcc | se sh cu (HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13])
keys = np.unique(cu)
cu_pace = np.array([pace[:, cu == k].mean() for k in keys])
cu_nl = np.array([nl[:, cu == k].mean() for k in keys])
within = np.mean([pace[:, cu == k].std(axis=1).mean() for k in keys])
print("CUs %d: pace per CU mean %.2f sd %.3f (max %.2f); sd of the pace WITHIN a CU %.3f; correlation CU pace vs CU planned agents %.3f" % (
    len(keys), cu_pace.mean(), cu_pace.std(), cu_pace.max(), within, np.corrcoef(cu_pace, cu_nl)[0, 1]))
tot_cu = np.array([tot[:, cu == k].mean(axis=1) for k in keys])       # [CUs, reps] mean total of the CU's workgroups
print("per-CU mean total: sd over CUs %.2f us, max - mean %.1f us; workgroup total minus its CU's mean: sd %.2f us" % (
    tot_cu.std(axis=0).mean(), (tot_cu.max(axis=0) - tot_cu.mean(axis=0)).mean(),
    np.mean([np.concatenate([tot[r, cu == k] - tot[r, cu == k].mean() for k in keys]).std() for r in range(tot.shape[0])])))
print("workgroups per CU: %s; blocks of the first CUs: %s" % (np.unique([int((cu == k).sum()) for k in keys], return_counts=True), [list(np.nonzero(cu == k)[0]) for k in keys[:3]]))
xcc = (keys >> 16) & 0xF
print("pace by XCD: " + " ".join("%d:%.2f" % (x, cu_pace[xcc == x].mean()) for x in np.unique(xcc)))
# the last finishers
r = d.shape[0] - 1
last = np.argsort(tot[r])[-12:]
print("the 12 last finishers of the last launch: total, planned agents, lp3 queue, pace first / second half")
for i in last:
    print("  wg %4d cu %05x total %.1f planned %.1f lp3 %.2f first %.2f second %.2f" % (i, cu[i], tot[r, i], nl[r, i], q3[r, i], first[r, i], second[r, i]))
