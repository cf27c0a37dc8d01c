#!/bin/bash
# Round 6, call j: the role priorities of the pipelined kernel (tuned in round 3 on single launches) re-checked under the
# n-step kernel with progress-fair priorities.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06j
mkdir -p $O
cd $R
G=$R/gym_collision_avoidance_amd
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-40s %-13s value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for v in fast "dPIPE_SPRIO4=1,fast" "dPIPE_SPRIO=2,dPIPE_SPRIO4=1,fast" "dPIPE_OPRIO=3,fast" "dPIPE_SPRIO=0,fast" "dPIPE_OPRIO=1,dPIPE_SPRIO=0,fast"; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --steps 20 --warmup 5 > "$O/l20_${v}_$rep.json" 2> "$O/l20_${v}_$rep.err"; show "$O/l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 2000 --mode rollout > "$O/ro_${v}_$rep.json" 2> "$O/ro_${v}_$rep.err"; show "$O/ro_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 500 --mode step > "$O/st_${v}_$rep.json" 2> "$O/st_${v}_$rep.err"; show "$O/st_${v}_$rep.json" "$v"
done
done
