#!/bin/bash
# Round 5, call f: ga3c_kernel scheduling variants (non-volatile gate asm; sched_group_barrier pipelines) against the product.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
export TMPDIR=/tmp
for lib in libcagpu_GA3C_NOVOL libcagpu_dGA3C_SGB=3; do
  CAGPU_LIB=$R/gym_collision_avoidance_amd/$lib.so timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "ga3c or checkpoint" > $O/tests_$lib.log 2>&1
  tail -2 $O/tests_$lib.log
done
for rep in 1 2; do
  for lib in libcagpu libcagpu_GA3C_NOVOL libcagpu_dGA3C_SGB=2 libcagpu_dGA3C_SGB=3; do
    CAGPU_LIB=$R/gym_collision_avoidance_amd/$lib.so timeout 300 python scratch/ga3c_rows.py > $O/rows_${lib}_$rep.json 2> $O/rows_${lib}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05f/rows_*.json")):
    try:
        d = json.load(open(f))
        print("%-44s" % f.split("/")[-1], "mean %.1f us;" % d["us_mean"], " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"][::2]))
    except Exception as e:
        print(f, "failed", e)
PY
