#!/usr/bin/env python3
"""scratch/ga3c_phases.py -- in-kernel phase timers of ga3c_kernel (GTICK slots; needs an -DCAGPU_ABLATE build:
python gym_collision_avoidance_amd/build_native.py ablate; CAGPU_LIB=.../libcagpu_ablate.so).  Cycles of thread 0 of every
tile between the marks, summed over the tiles of N launches on the config-3 workload."""
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from gym_collision_avoidance_amd import _native as nat  # noqa: E402

NAMES = ["prologue (rows, X, normalise)", "LSTM weights + barrier", "LSTM (<= 19 steps)", "layer1", "layer2", "fc1", "logits"]
dev = torch.device("cuda", 0)
sim, table, N, K = bench.build_workload("ga3c20", 4096, dev)
for _ in range(int(os.environ.get("WARM", "30"))):
    sim.step()
lib = nat.lib()
buf = (C.c_ulonglong * 16)()
torch.cuda.synchronize()
lib.cagpu_debug_prof(buf, 1)
n = 20
for _ in range(n):
    sim.ga3c()
torch.cuda.synchronize()
lib.cagpu_debug_prof(buf, 0)
rows = sim.ga3c_rows()
tot = sum(buf[i] for i in range(7)) + buf[14] + buf[15]
print("rows %d; cycles per launch summed over tiles, share" % rows)
for i, nm in enumerate(NAMES):
    print("  %-32s %12.0f  %5.1f %%" % (nm, buf[i] / n, 100.0 * buf[i] / tot))
print("  per 64-row tile equivalent: %.0f cycles" % (tot / n / (rows / 64.0)))
if buf[14]:
    print("  prologue in parts: start -> loads issued %.0f, -> first barrier passed %.0f, normalise + store %.0f (per launch, summed over tiles)"
          % (buf[14] / n, buf[15] / n, buf[0] / n))
if buf[12]:
    if os.environ.get("FLOW_STAGE"):   # lstm_flow: stamps inside the stage of row block 1
        SEG = ["-", "stage top -> first MFMA issued", "-> MFMA 8 issued", "-> MFMA 32 issued", "-> MFMA 56 issued"]
    else:
        SEG = ["MFMA stages (+ cell updates between)", "last row block's cell update", "wait at barrier 1", "split + store h", "wait at barrier 2"]
    print("LSTM step of wave 0, cycles per step (mean over %d tile-steps, %.2f row blocks per tile):" % (buf[12] / n, buf[13] / buf[12]))
    for k, nm in enumerate(SEG):
        print("  %-40s %8.0f" % (nm, buf[7 + k] / buf[12]))
    print("  %-40s %8.0f" % ("sum", sum(buf[7 + k] for k in range(5)) / buf[12]))
