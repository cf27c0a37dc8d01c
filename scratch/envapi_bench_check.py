import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
for i in range(3):
    r = bench.env_api_rates(4096, 10, 640, torch, dev)
    print({k: (round(v["us_per_step"], 2), v.get("ring")) for k, v in r.items() if isinstance(v, dict)})
