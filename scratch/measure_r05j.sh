#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "laser or config5 or crowd or scan or map" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for rep in 1 2; do
timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_$rep.json 2> $O/cfg5.err
python - $O/cfg5_$rep.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("cfg5 value %.3e ms/step %.4f step kernel %.1f us scan kernel %.1f us frac %.4f" % (d["value"], d["ms_per_step"], r["step_kernel_us"], r["scan_kernel_us"], r["frac"]))
PY
done
