#!/bin/bash
# round 4, call C: the clearance-grid laser march: parity (laserscan tests, config-5 geometry test), then A/B of the config-5
# bench line against the -DCAGPU_SCAN_DT=0 build (round 3's march), two repetitions each; the batched host-policy test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_api.py tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider --timeout 600 -k "laser or config5 or host_path or host_fallback" > $O/laser.log 2>&1
echo "laser rc=$?" >> $O/laser.log
tail -n 30 $O/laser.log
for rep in 1 2; do
  timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/crowd_dt1_$rep.json 2>$O/crowd_dt1_$rep.err
  CAGPU_LIB="$PWD/gym_collision_avoidance_amd/libcagpu_dSCAN_DT=0.so" timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/crowd_dt0_$rep.json 2>$O/crowd_dt0_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04c/crowd*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print("%-22s step+scan %.3f ms  (step %.1f us, scan %.1f us)  frac %.4f" % (f.split("/")[-1], d["ms_per_step"], r["step_kernel_us"], r["scan_kernel_us"], r["frac"]))
    except Exception as e: print(f, "failed", e)
PY
