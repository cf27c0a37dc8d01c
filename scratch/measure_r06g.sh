#!/bin/bash
# Round 6, call g: how much of the n-step kernel's VALU issue goes into POLLS (an LDS read + a v_readfirstlane per poll of a
# hand-over counter)?  Longer sleeps on the waits that are not on a step's critical chain (-DCAGPU_PIPE_NCSLEEP=2/4/8), same box:
# time per step + the VALU instruction counters of the 50-step launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06g
mkdir -p $O
cd $R
export TMPDIR=/tmp
G=$R/gym_collision_avoidance_amd
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s %-13s E %5d value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["config"]["envs_per_gpu"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for v in fast "dPIPE_NCSLEEP=2,fast" "dPIPE_NCSLEEP=4,fast" "dPIPE_NCSLEEP=8,fast"; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --steps 20 --warmup 5 > "$O/l20_${v}_$rep.json" 2> "$O/l20_${v}_$rep.err"; show "$O/l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 200 --lookahead 50 > "$O/l50_${v}_$rep.json" 2> "$O/l50_${v}_$rep.err"; show "$O/l50_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 2000 --mode rollout > "$O/ro_${v}_$rep.json" 2> "$O/ro_${v}_$rep.err"; show "$O/ro_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 500 --mode step > "$O/st_${v}_$rep.json" 2> "$O/st_${v}_$rep.err"; show "$O/st_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 1024 --steps 640 > "$O/e1024_${v}_$rep.json" 2> "$O/e1024_${v}_$rep.err"; show "$O/e1024_${v}_$rep.json" "$v"
done
done
cd /tmp
P="python $R/bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0 --steps 500 --lookahead 50 --warmup 500 --min-warm-seconds 0"
for v in fast "dPIPE_NCSLEEP=4,fast"; do
  CAGPU_LIB=$G/libcagpu_$v.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$O/pmc_$v" -- $P > "$O/pmc_$v.log" 2>&1
  CAGPU_LIB=$G/libcagpu_$v.so timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --output-format csv -d "$O/pmc2_$v" -- $P > "$O/pmc2_$v.log" 2>&1
done
find $O -name '*agent_info.csv' -delete
cd $R
python profiles/summarize.py "$O/pmc_fast" "$O/pmc_dPIPE_NCSLEEP=4,fast" "$O/pmc2_fast" "$O/pmc2_dPIPE_NCSLEEP=4,fast" | grep -E "ca_pipe_kernel<10, 4, true" 
