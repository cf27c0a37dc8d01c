#!/bin/bash
# config 5 (bench.py --workload crowd50_laser) and config 3, the product against libcagpu_prev.so on the same box, after the
# sensing / sorting / crowd parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/cfg5_ab
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "closest or sort or config or fuzzed or ragged or crowd or big or laser or tie" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -n 3 $O/tests.log
PREV=$PWD/gym_collision_avoidance_amd/libcagpu_prev.so
for rep in 1 2; do
  timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_new_$rep.json 2> $O/cfg5_new_$rep.err
  CAGPU_LIB=$PREV timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_prev_$rep.json 2> $O/cfg5_prev_$rep.err
done
timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_new_1.json 2> $O/cfg3_new_1.err
CAGPU_LIB=$PREV timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_prev_1.json 2> $O/cfg3_prev_1.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg5_ab/cfg*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print("%-22s" % f.split("/")[-1], "ms_per_step %.4f value %.3e" % (d["ms_per_step"], d["value"]), "step %.1f scan %.1f" % (r.get("step_kernel_us", 0), r.get("scan_kernel_us", 0)))
    except Exception as e:
        print(f, "failed", e)
PY
