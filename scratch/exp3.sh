#!/bin/bash
# scratch/exp3.sh -- iteration loop: GPU tests (fail fast), phase profile (ablate build), bench (product build)
R=$PWD
O=$R/gpurun_out/${TAG:-exp3}
mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/test_gpu.log 2>&1
tail -25 $O/test_gpu.log
AB=$R/gym_collision_avoidance_amd/libcagpu_ablate_fast.so
CAGPU_LIB=$AB timeout 300 python scratch/prof_phases.py > $O/phases_step.txt 2>&1
CAGPU_LIB=$AB MODE=rollout timeout 300 python scratch/prof_phases.py > $O/phases_rollout.txt 2>&1
cat $O/phases_step.txt; tail -8 $O/phases_rollout.txt
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_driver.json 2>> $O/bench.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.3e" % d["value"], "wall ms/step %.4f" % d["ms_per_step"], "event %.4f" % d.get("event_ms_per_step", -1),
              "suspect", d.get("suspect"), "rollout %.4f" % d.get("rollout", {}).get("ms_per_step", -1),
              "2streams %.4f" % d.get("two_streams", {}).get("ms_per_step", -1), d["roofline"]["kernel"][:60])
    except Exception as e:
        print(f, "FAILED", e, open("$O/bench.err").read()[-600:])
PY
