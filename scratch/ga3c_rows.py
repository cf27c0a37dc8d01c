#!/usr/bin/env python3
"""scratch/ga3c_rows.py -- why does ga3c_kernel take 304 us at one state and 372 us on average under the profiler (VERDICT
r03 weak-7)?  Steps the config-3 workload and records, per step, the number of rows cagpu_ga3c evaluated (live GA3C-CADRL
agents, read back from the packing's counter) and the device time of that launch (HIP events)."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
sim, table, N, K = bench.build_workload("ga3c20", 4096, dev)
for _ in range(30):
    sim.step()
rec = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for t in range(240):
    torch.cuda.synchronize(dev)
    e0.record()
    sim.ga3c()
    e1.record()
    torch.cuda.synchronize(dev)
    rec.append((sim.ga3c_rows(), e0.elapsed_time(e1) * 1e3))
    sim.step()
rows = np.array([r for r, _ in rec], float)
us = np.array([u for _, u in rec], float)
fit = np.polyfit(rows, us, 1)
out = {"steps": len(rec), "rows_min": rows.min(), "rows_mean": rows.mean(), "rows_max": rows.max(),
       "us_min": us.min(), "us_mean": us.mean(), "us_max": us.max(),
       "linear_fit_us": {"per_1000_rows": fit[0] * 1e3, "intercept": fit[1]},
       "corr_rows_us": float(np.corrcoef(rows, us)[0, 1]),
       "by_rows": [{"rows": int(r), "us": round(u, 1)} for r, u in sorted(rec)[::12]],
       "note": "compaction + network of one cagpu_ga3c call, HIP events; the launch time follows the number of live rows "
               "(tile rounds of 512 resident workgroups), not the clock or the profiler"}
print(json.dumps(out))
