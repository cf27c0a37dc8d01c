#!/bin/bash
# round 4, call F: the network on the bf16 matrix cores (three-plane split operands): parity tests, launch time by rows, config 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04g
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "ga3c or checkpoint" > $O/ga3c_tests.log 2>&1
echo "tests rc=$?" >> $O/ga3c_tests.log
tail -n 25 $O/ga3c_tests.log
for rep in 1 2; do
  timeout 300 python scratch/ga3c_rows.py > $O/rows_$rep.json 2> $O/rows_$rep.err
  CAGPU_LIB=$PWD/gym_collision_avoidance_amd/libcagpu_splitread.so timeout 300 python scratch/ga3c_rows.py > $O/rows_splitread_$rep.json 2> $O/rows_splitread_$rep.err
  timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_$rep.json 2> $O/cfg3_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g/rows_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "mean %.1f us at %.0f rows;" % (d["us_mean"], d["rows_mean"]),
              " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"]))
    except Exception as e:
        print(f, "failed", e)
for f in sorted(glob.glob("gpurun_out/r04g/cfg3_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "ms_per_step %.4f value %.3e net %.1f us frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]), d["timed_blocks"])
    except Exception as e:
        print(f, "failed", e)
PY
