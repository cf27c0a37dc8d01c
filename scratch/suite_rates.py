import os, sys
sys.path.insert(0, os.getcwd())
os.environ["GYM_CONFIG_PATH"] = os.path.join(os.getcwd(), "tests", "env_configs.py")
os.environ["GYM_CONFIG_CLASS"] = "FullTestSuite"
import numpy as np
from gym_collision_avoidance_amd.experiments import run_full_test_suite as suite
import time
for pol in ("GA3C-CADRL-10", "RVO"):
    for n in (2, 3, 4, 6, 8, 10):
        t0 = time.time()
        df = suite.run_suite(pol, n)
        ok = df[df["all_at_goal"]]
        print("%-14s N=%2d cases=%d all_at_goal=%.1f%% collision=%.1f%% stuck=%.1f%% mean_extra_time(success)=%.2fs  [%.1fs]" % (
            pol, n, len(df), 100 * df["all_at_goal"].mean(), 100 * df["collision"].mean(),
            100 * (df["outcome"] == "stuck").mean(),
            np.mean([np.mean(x) for x in ok["extra_time_to_goal"]]) if len(ok) else float("nan"), time.time() - t0), flush=True)
