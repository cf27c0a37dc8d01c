#!/bin/bash
# scratch/exp2.sh -- phase profile of the step / rollout kernels (ablate build), driver-style bench, GPU tests
R=$PWD
O=$R/gpurun_out/exp2
mkdir -p $O
AB=$R/gym_collision_avoidance_amd/libcagpu_ablate_fast.so
CAGPU_LIB=$AB timeout 300 python scratch/prof_phases.py > $O/phases_step.txt 2>&1
CAGPU_LIB=$AB MODE=rollout timeout 300 python scratch/prof_phases.py > $O/phases_rollout.txt 2>&1
cat $O/phases_step.txt; tail -22 $O/phases_rollout.txt
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_driver$i.json 2>> $O/bench_driver.err; done
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/exp2/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.3e" % d["value"], "wall ms/step %.4f" % d["ms_per_step"], "event %.4f" % d.get("event_ms_per_step", -1),
              "suspect", d.get("suspect"), "rollout %.4f" % d.get("rollout", {}).get("ms_per_step", -1),
              "2streams %.4f" % d.get("two_streams", {}).get("ms_per_step", -1), d["roofline"]["kernel"][:60])
        if "cpu_baseline" in d: print("  cpu_baseline", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
    except Exception as e:
        print(f, "FAILED", e)
PY
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/test_gpu.log 2>&1
tail -5 $O/test_gpu.log
