#!/bin/bash
# Round 6, call b: the GPU suite on the build with progress-fair priorities + the async fault probe + the GA3C range guard;
# the driver-shaped line; 1024 x 10 (configs[1]) with 4- / 2- / 1-env tiles in ring mode (same box).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
cut -c1-600 $O/bench_driver.json
G=$R/gym_collision_avoidance_amd
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-34s %-13s E %5d value %.3e wall us/step %.3f events us/step %.3f  %s" % (sys.argv[2], d["config"]["launch_mode"], d["config"]["envs_per_gpu"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3, d["roofline"]["kernel"][:40]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for v in fast "dPIPE_SMALLTILE=2,fast" "dPIPE_SMALLTILE=1,fast"; do
  L=$G/libcagpu_$v.so
  CAGPU_LIB=$L timeout 120 $B --envs 1024 --steps 640 > "$O/e1024_l64_${v}_$rep.json" 2> "$O/e1024_l64_${v}_$rep.err"; show "$O/e1024_l64_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 1024 --steps 20 --warmup 5 > "$O/e1024_l20_${v}_$rep.json" 2> "$O/e1024_l20_${v}_$rep.err"; show "$O/e1024_l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 1024 --mode step --steps 500 > "$O/e1024_step_${v}_$rep.json" 2> "$O/e1024_step_${v}_$rep.err"; show "$O/e1024_step_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 1024 --mode rollout --steps 2000 > "$O/e1024_ro_${v}_$rep.json" 2> "$O/e1024_ro_${v}_$rep.err"; show "$O/e1024_ro_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 2048 --steps 640 > "$O/e2048_l64_${v}_$rep.json" 2> "$O/e2048_l64_${v}_$rep.err"; show "$O/e2048_l64_${v}_$rep.json" "$v"
done
done
for v in "dPIPE_SMALLTILE=2,fast" "dPIPE_SMALLTILE=1,fast"; do
  CAGPU_LIB=$G/libcagpu_$v.so timeout 600 python -m pytest tests/test_gpu_bench_geometry.py tests/test_gpu_ring.py -m gpu -q -p no:cacheprovider -k "1024" > "$O/pytest_$v.log" 2>&1; tail -3 "$O/pytest_$v.log"
done
