#!/bin/bash
# Round 6, call a: baseline GPU suite on the round's first build; packed-f32 issue rates (valu_rates); the progress-fair
# priority experiment (-DCAGPU_PIPE_YIELD=<T>[, -DCAGPU_PIPE_YLEVEL]) against the same-box baseline at the driver's shape
# (ring of 20), ring of 50 and the long rollout; per-step stamps with and without it.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 120 scratch/valu_rates > $O/valu_rates.txt 2>&1
grep -E "waves per SIMD|fma32|pk_|mul32|fma64" $O/valu_rates.txt
G=$R/gym_collision_avoidance_amd
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-44s %-13s value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for v in fast "dPIPE_YIELD=1,fast" "dPIPE_YIELD=2,fast" "dPIPE_YIELD=3,fast" "dPIPE_YIELD=1,dPIPE_YLEVEL=1,fast" "dPIPE_YIELD=2,dPIPE_YLEVEL=1,fast"; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --steps 20 --warmup 5 > "$O/l20_${v}_$rep.json" 2> "$O/l20_${v}_$rep.err"; show "$O/l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 200 --lookahead 50 > "$O/l50_${v}_$rep.json" 2> "$O/l50_${v}_$rep.err"; show "$O/l50_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 2000 --mode rollout > "$O/ro_${v}_$rep.json" 2> "$O/ro_${v}_$rep.err"; show "$O/ro_${v}_$rep.json" "$v"
done
done
for v in "steptime,fast" "steptime,dPIPE_YIELD=1,fast" "steptime,dPIPE_YIELD=2,fast"; do
  echo "==== $v" | tee -a $O/steptime.txt
  CAGPU_LIB=$G/libcagpu_$v.so timeout 200 python scratch/steptime.py 20 >> $O/steptime.txt 2>&1
done
grep -E "====|launch span|per-workgroup total|per-CU mean total|persistence" $O/steptime.txt
