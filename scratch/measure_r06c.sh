#!/bin/bash
# Round 6, call c: same-box A/B of the product's progress-fair priorities (launcher-set yield_t = 2, level 1) against the same
# code with -DCAGPU_PIPE_YIELD_T=0, per-step stamps of both; the new ring / RCCL tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_rccl.py -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_ring.log 2>&1; echo "rc=$?" >> $O/pytest_ring.log
tail -5 $O/pytest_ring.log | cut -c1-300
G=$R/gym_collision_avoidance_amd
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
for rep in 1 2 3; do
for v in "dPIPE_YIELD_T=0,fast" fast; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --steps 20 --warmup 5 > "$O/l20_${v}_$rep.json" 2> "$O/l20_${v}_$rep.err"; show "$O/l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 200 --lookahead 50 > "$O/l50_${v}_$rep.json" 2> "$O/l50_${v}_$rep.err"; show "$O/l50_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 2000 --mode rollout > "$O/ro_${v}_$rep.json" 2> "$O/ro_${v}_$rep.err"; show "$O/ro_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 2048 --steps 640 > "$O/e2048_${v}_$rep.json" 2> "$O/e2048_${v}_$rep.err"; show "$O/e2048_${v}_$rep.json" "$v"
done
done
for v in "steptime,dPIPE_YIELD_T=0,fast" "steptime,fast"; do
  echo "==== $v" | tee -a $O/steptime.txt
  CAGPU_LIB=$G/libcagpu_$v.so timeout 200 python scratch/steptime.py 20 >> $O/steptime.txt 2>&1
done
grep -E "====|launch span|per-workgroup total|per-CU mean total|persistence" $O/steptime.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
show $O/bench_driver.json product
