#!/bin/bash
# round 4, call H: LSTM gates with one reciprocal per sigmoid * tanh pair (libcagpu_gates.so) against the product, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
G=$PWD/gym_collision_avoidance_amd/libcagpu_gates.so
CAGPU_LIB=$G timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "ga3c or checkpoint" > $O/ga3c_tests.log 2>&1
echo "tests rc=$?" >> $O/ga3c_tests.log
tail -n 5 $O/ga3c_tests.log
for rep in 1 2; do
  timeout 300 python scratch/ga3c_rows.py > $O/rows_prod_$rep.json 2> $O/rows_prod_$rep.err
  CAGPU_LIB=$G timeout 300 python scratch/ga3c_rows.py > $O/rows_gates_$rep.json 2> $O/rows_gates_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04h/rows_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "mean %.1f us at %.0f rows;" % (d["us_mean"], d["rows_mean"]),
              " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"]))
    except Exception as e:
        print(f, "failed", e)
PY
