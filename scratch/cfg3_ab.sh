#!/bin/bash
# config 3 (bench.py --workload ga3c20), the product against libcagpu_prev.so on the same box, after the GA3C / sort tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/cfg3_ab
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "ga3c or checkpoint or closest or sort or config3" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -n 3 $O/tests.log
PREV=$PWD/gym_collision_avoidance_amd/libcagpu_prev.so
for rep in 1 2; do
  timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_new_$rep.json 2> $O/cfg3_new_$rep.err
  CAGPU_LIB=$PREV timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_prev_$rep.json 2> $O/cfg3_prev_$rep.err
done
timeout 300 python bench.py --workload ga3c20 --ga3c-fused --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_fused_new_1.json 2> $O/cfg3_fused_new_1.err
CAGPU_LIB=$PREV timeout 300 python bench.py --workload ga3c20 --ga3c-fused --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_fused_prev_1.json 2> $O/cfg3_fused_prev_1.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg3_ab/cfg3_*.json")):
    try:
        d = json.load(open(f))
        print("%-28s" % f.split("/")[-1], "ms_per_step %.4f value %.3e net %.1f us rows %d" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["rows_evaluated"]), "blocks min %.4f max %.4f" % (d["timed_blocks"]["ms_per_step_min"], d["timed_blocks"]["ms_per_step_max"]))
    except Exception as e:
        print(f, "failed", e)
PY
