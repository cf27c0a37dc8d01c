#!/bin/bash
# same-box A/B of ga3c_kernel builds: scratch/ga3c_rows.py for every library named (3 repetitions, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/ga3c_ab3
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for lib in "$@"; do
    n=$(basename "$lib" .so)
    CAGPU_LIB="$PWD/$lib" timeout 300 python scratch/ga3c_rows.py > "$O/rows_${n}_$rep.json" 2> "$O/rows_${n}_$rep.err"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ga3c_ab3/rows_*.json")):
    try:
        d = json.load(open(f))
        print("%-44s" % f.split("/")[-1], "mean %.1f us;" % d["us_mean"], " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"][::3]))
    except Exception as e:
        print(f, "failed", e)
PY
