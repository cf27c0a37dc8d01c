#!/usr/bin/env python3
"""scratch/ga3c_loop.py -- N back-to-back cagpu_ga3c calls on the config-3 workload after WARM steps (for rocprofv3 passes)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
sim, table, N, K = bench.build_workload("ga3c20", 4096, dev)
for _ in range(int(os.environ.get("WARM", "30"))):
    sim.step()
torch.cuda.synchronize()
for _ in range(int(os.environ.get("N", "20"))):
    sim.ga3c()
torch.cuda.synchronize()
print("rows", sim.ga3c_rows())
