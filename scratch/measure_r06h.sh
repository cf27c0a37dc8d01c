#!/bin/bash
# Round 6, call h: the single-step kernel (what a caller with external actions gets) with a STATIC priority bonus for the
# workgroups dispatched last onto a CU (-DCAGPU_PIPE_YOUNG=1: last quarter of the grid, 2: last half), same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
export TMPDIR=/tmp
G=$R/gym_collision_avoidance_amd
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-24s %-13s E %5d value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["config"]["envs_per_gpu"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do
for v in fast "dPIPE_YOUNG=1,fast" "dPIPE_YOUNG=2,fast"; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --steps 500 --mode step > "$O/st_${v}_$rep.json" 2> "$O/st_${v}_$rep.err"; show "$O/st_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 500 --mode graph > "$O/gr_${v}_$rep.json" 2> "$O/gr_${v}_$rep.err"; show "$O/gr_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 300 --mode step --envs 2048 > "$O/st2048_${v}_$rep.json" 2> "$O/st2048_${v}_$rep.err"; show "$O/st2048_${v}_$rep.json" "$v"
done
done
