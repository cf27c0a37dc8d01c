#!/bin/bash
# scratch/exp1.sh -- round-2 GPU call 1: wave placement probe, agent-wave rotation sweep, driver-style bench, GPU tests.
R=$PWD
O=$R/gpurun_out/exp1
mkdir -p $O
KN=$R/gym_collision_avoidance_amd/libcagpu_knobs.so
timeout 60 scratch/hwid 1024 23552 4000 > $O/hwid.txt 2>&1
timeout 60 scratch/hwid 768 41104 4000 > $O/hwid_staged.txt 2>&1
head -8 $O/hwid.txt; tail -14 $O/hwid.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver2.json 2>> $O/bench_driver.err
for aw in -1 0 1 2 3 4 5 6 8; do
  CAGPU_LIB=$KN CAGPU_AW_SHIFT=$aw timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/aw_$aw.json 2> $O/aw_$aw.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/exp1/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.3e" % d["value"], "wall ms/step %.4f" % d["ms_per_step"], "event %.4f" % d.get("event_ms_per_step", -1),
              "suspect", d.get("suspect"), "rollout %.4f" % d.get("rollout", {}).get("ms_per_step", -1),
              "2streams %.4f" % d.get("two_streams", {}).get("ms_per_step", -1), d["roofline"]["kernel"][:60])
    except Exception as e:
        print(f, "FAILED", e)
PY
python - <<'PY' > $O/hostfloor.txt 2>&1
import time, torch, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from gym_collision_avoidance_amd import _native as nat, core
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n10"]
for E in (8, 4096):
    sim = core.BatchedSim(core.make_params(E, 10)); sim.set_plugins(nat.POL_RVO); sim.set_fixture_table(table); sim.reset_from_table()
    for _ in range(300): sim.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): sim.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("E", E, "host enqueue us/launch %.2f" % ((t1 - t0) / 3000 * 1e6), "incl. drain %.2f" % ((t2 - t0) / 3000 * 1e6))
PY
cat $O/hostfloor.txt
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/test_gpu.log 2>&1
tail -5 $O/test_gpu.log
