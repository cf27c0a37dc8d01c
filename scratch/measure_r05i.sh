#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for V in "" "HSA_ENABLE_INTERRUPT=0"; do
  env $V timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/x.json 2> $O/x.err
  python - "$O/x.json" "$V" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-24s %-13s value %.3e wall us/step %.3f events %.3f" % (sys.argv[2] or "default", d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
PY
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_geometry.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
