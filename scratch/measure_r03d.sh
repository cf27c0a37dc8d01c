#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "pipelined or orca_velocities or metric_geometry or rollout_equals or suite" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 3000"
for rep in 1 2; do
timeout 200 $B > $O/b_pipe_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03d/b_*.json")):
    try:
        d=json.load(open(f)); print("%-40s step %.2f us  rollout %.2f us/step" % (f.split("/")[-1], d["event_ms_per_step"]*1e3, d.get("rollout",{}).get("ms_per_step",0)*1e3))
    except Exception as e: print(f, "failed", e)
PY
CAGPU_LIB=gym_collision_avoidance_amd/libcagpu_pipetime_fast.so timeout 300 python scratch/pipetime.py > $O/pipetime.txt 2>&1; cat $O/pipetime.txt
