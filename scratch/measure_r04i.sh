#!/bin/bash
# round 4, call I: closest_last without its second ranking pass (ca_kernel P4, fused GA3C prologue) + the epoch-tagged counters
# of compact_kernel (no memset per call): the whole GPU suite, then config 3 against the previous library on the same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04i
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 5 $O/pytest_gpu.log
PREV=$PWD/gym_collision_avoidance_amd/libcagpu_prev.so
for rep in 1 2; do
  timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_new_$rep.json 2> $O/cfg3_new_$rep.err
  CAGPU_LIB=$PREV timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_prev_$rep.json 2> $O/cfg3_prev_$rep.err
  timeout 300 python bench.py --workload ga3c20 --ga3c-fused --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_fused_new_$rep.json 2> $O/cfg3_fused_new_$rep.err
  CAGPU_LIB=$PREV timeout 300 python bench.py --workload ga3c20 --ga3c-fused --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_fused_prev_$rep.json 2> $O/cfg3_fused_prev_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04i/cfg3_*.json")):
    try:
        d = json.load(open(f))
        print("%-28s" % f.split("/")[-1], "ms_per_step %.4f value %.3e net %.1f us" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"]), "blocks min %.4f max %.4f" % (d["timed_blocks"]["ms_per_step_min"], d["timed_blocks"]["ms_per_step_max"]))
    except Exception as e:
        print(f, "failed", e)
PY
