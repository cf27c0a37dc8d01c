#!/usr/bin/env python3
"""How much of a launch is load imbalance between tiles?  A probe BEFORE building the env -> tile table into the kernels:
the envs of the metric batch are PHYSICALLY re-ordered in memory before every timed launch so that every 4-env tile holds
about the same number of planned (ORCA-querying) agents (envs sorted by that count, dealt to the tiles in snake order), and
the launch is timed against the natural order (tile = 4 consecutive envs = a random draw of heavy and light envs).  The
re-ordering changes which fixture case an env loads at its next auto-reset (the case index follows the env id) -- irrelevant
for a timing probe, which is all this is.  usage: balance_probe.py [E]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from gym_collision_avoidance_amd import _native as nat  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
sim, table, N, K = bench.build_workload("rvo10", E, dev)
sim.rollout(300)
torch.cuda.synchronize()
T = (E + 3) // 4


def planned_counts():
    f = sim.state["flags"]
    stop = nat.AT_GOAL | nat.OUT_OF_TIME | nat.IN_COLLISION | nat.ABSENT
    return ((f & stop) == 0).sum(dim=1)


def balance(mode):
    cnt = planned_counts()
    if mode == "sorted":        # the worst case: heavy envs together
        order = torch.argsort(cnt, descending=True, stable=True)
    else:                        # snake dealing: tile t gets ranks t, 2T-1-t, 2T+t, 4T-1-t
        srt = torch.argsort(cnt, descending=True, stable=True)
        r = torch.arange(E, device=dev)
        tile, j = r // 4, r % 4
        rank = torch.where(j % 2 == 0, j * T + tile, (j + 1) * T - 1 - tile)
        order = srt[rank.clamp(max=E - 1)]
    for n, t in sim._state.items():
        t.copy_(t[order])
    return cnt[order].view(-1, 4).sum(dim=1).float()


def timed(launch, n, mode):
    tot, spread = 0.0, []
    for _ in range(n):
        if mode != "natural":
            spread.append(balance(mode).std().item())
        else:
            spread.append(planned_counts().view(-1, 4).sum(dim=1).float().std().item())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3, float(np.mean(spread))


for mode in ("natural", "balanced", "sorted"):
    for L in (1, 20, 50):
        launch = (lambda: sim.step()) if L == 1 else (lambda L=L: sim.rollout(L))
        for _ in range(10):
            launch()
        us, sd = timed(launch, 60 if L == 1 else 30, mode)
        print("E %d L %3d %-9s %8.2f us / launch = %6.3f us / step   (planned agents per tile: sd %.2f)" % (E, L, mode, us, us / L, sd))
