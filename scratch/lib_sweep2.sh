#!/bin/bash
# A/B of build variants of the step kernel in ONE gpurun call (boxes differ): every libcagpu*.so named on the command line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/sweep
mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so)
  CAGPU_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --steps 3000 > $O/${n}_$rep.json 2>/dev/null
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/sweep/*.json")):
    try:
        d=json.load(open(f)); print("%-60s step %.2f us  rollout %.2f us/step" % (f.split("/")[-1], d["event_ms_per_step"]*1e3, d.get("rollout",{}).get("ms_per_step",0)*1e3))
    except Exception as e: print(f, "failed", e)
PY
