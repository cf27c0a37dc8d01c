#!/bin/bash
# A/B of ga3c_kernel builds in ONE gpurun call: the product (after its GA3C tests) against every libcagpu_*.so named on the
# command line; scratch/ga3c_rows.py = launch time by live rows on the config-3 workload, two repetitions each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/ga3c_ab
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "ga3c or checkpoint" > $O/ga3c_tests.log 2>&1
echo "tests rc=$?" >> $O/ga3c_tests.log
tail -n 4 $O/ga3c_tests.log
for rep in 1 2; do
  timeout 300 python scratch/ga3c_rows.py > $O/rows_product_$rep.json 2> $O/rows_product_$rep.err
  for lib in "$@"; do
    n=$(basename $lib .so)
    CAGPU_LIB=$PWD/$lib timeout 300 python scratch/ga3c_rows.py > $O/rows_${n}_$rep.json 2> $O/rows_${n}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ga3c_ab/rows_*.json")):
    try:
        d = json.load(open(f))
        print("%-34s" % f.split("/")[-1], "mean %.1f us;" % d["us_mean"], " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"][::2]))
    except Exception as e:
        print(f, "failed", e)
PY
