#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider --timeout 900 -k "episode_outcomes" -s ) > $O/outcome_test.log 2>&1
tail -8 $O/outcome_test.log
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.3"
for rep in 1 2; do
for lib in fast dPIPE_TILEPRIO=1,fast; do
  for M in "--steps 20 --lookahead 20 --warmup 5" "--steps 500 --lookahead 50" "--mode step --steps 500" "--mode rollout --steps 1000"; do
    CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_$lib.so timeout 120 $B $M > $O/x.json 2> $O/x.err
    python - "$O/x.json" "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-24s %-13s value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
PY
  done
done
done
