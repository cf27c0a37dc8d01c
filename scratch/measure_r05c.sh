#!/bin/bash
# Round 5, call c: wave placement of the 8-wave workgroups; staggered workgroup starts (dPIPE_STAGGER builds) against the
# plain fast build, lookahead 20 / 50 and one launch per step, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
export TMPDIR=/tmp
scratch/hwid8 > $O/hwid8.txt 2>&1
cat $O/hwid8.txt
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.3"
for rep in 1 2; do
for lib in fast dPIPE_STAGGER=2,fast dPIPE_STAGGER=4,fast dPIPE_STAGGER=8,fast dPIPE_STAGGER=-4,fast; do
  for M in "--steps 200 --lookahead 20" "--steps 500 --lookahead 50" "--mode step --steps 500" "--mode rollout --steps 1000"; do
    CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_$lib.so timeout 120 $B $M > $O/x.json 2> $O/x.err
    python - "$O/x.json" "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-24s %-13s value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
PY
  done
done
done
