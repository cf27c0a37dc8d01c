#!/bin/bash
# round 4, call E: balanced per-tile heights of ga3c_kernel against the one-height-per-launch policy, same box.
# libcagpu_oldtile.so = the previous commit's policy built as a variant (CAGPU_LIB).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04e
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -k "ga3c or checkpoint" > $O/ga3c_tests.log 2>&1
echo "tests rc=$?" >> $O/ga3c_tests.log
tail -n 6 $O/ga3c_tests.log
OLD=$PWD/gym_collision_avoidance_amd/libcagpu_oldtile.so
for rep in 1 2; do
  timeout 300 python scratch/ga3c_rows.py > $O/rows_new_$rep.json 2> $O/rows_new_$rep.err
  CAGPU_LIB=$OLD timeout 300 python scratch/ga3c_rows.py > $O/rows_old_$rep.json 2> $O/rows_old_$rep.err
  timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_new_$rep.json 2> $O/cfg3_new_$rep.err
  CAGPU_LIB=$OLD timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_old_$rep.json 2> $O/cfg3_old_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04e/rows_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "mean %.1f us at %.0f rows;" % (d["us_mean"], d["rows_mean"]),
              " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"]))
    except Exception as e:
        print(f, "failed", e)
for f in sorted(glob.glob("gpurun_out/r04e/cfg3_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "ms_per_step %.4f value %.3e frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
    except Exception as e:
        print(f, "failed", e)
PY
