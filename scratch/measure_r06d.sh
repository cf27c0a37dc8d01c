#!/bin/bash
# Round 6, call d: (1) CU-balanced dealing of the tiles (block -> tile table by planned-agent count, probed from the host:
# `tilemap` build + scratch/steptime.py BALANCE=cu) against the natural order, in-kernel stamps; (2) the round-5 end state
# against this round's build on the same box; (3) yield level / threshold variants.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06d
mkdir -p $O
cd $R
export TMPDIR=/tmp
G=$R/gym_collision_avoidance_amd
for L in 20 50; do
for mode in "" cu; do
  echo "==== tilemap L=$L BALANCE=$mode" | tee -a $O/steptime.txt
  BALANCE=$mode CAGPU_LIB="$G/libcagpu_steptime,tilemap,fast.so" timeout 200 python scratch/steptime.py $L >> $O/steptime.txt 2>&1
done
done
grep -E "====|launch span|per-workgroup total|per-CU mean total|persistence|CUs 256" $O/steptime.txt
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-34s %-13s E %5d value %.3e wall us/step %.3f events us/step %.3f  %s" % (sys.argv[2], d["config"]["launch_mode"], d["config"]["envs_per_gpu"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3, d["roofline"]["kernel"][-24:]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for v in r05end_fast fast "dPIPE_YLEVEL=0,fast" "dPIPE_YIELD_T=3,fast" "dPIPE_YLEVEL=2,fast"; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --steps 20 --warmup 5 > "$O/l20_${v}_$rep.json" 2> "$O/l20_${v}_$rep.err"; show "$O/l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 200 --lookahead 50 > "$O/l50_${v}_$rep.json" 2> "$O/l50_${v}_$rep.err"; show "$O/l50_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 2000 --mode rollout > "$O/ro_${v}_$rep.json" 2> "$O/ro_${v}_$rep.err"; show "$O/ro_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 500 --mode step > "$O/st_${v}_$rep.json" 2> "$O/st_${v}_$rep.err"; show "$O/st_${v}_$rep.json" "$v"
done
done
