#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ring.py tests/test_gpu_rccl.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/pytest_ring.log 2>&1; echo "rc=$?" >> $O/pytest_ring.log
tail -12 $O/pytest_ring.log
for rep in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_driver_$rep.json 2> $O/bench_driver.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_n1_$rep.json 2> $O/bench_n1.err
for f in bench_driver_$rep bench_n1_$rep; do python - $O/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-18s %-13s value %.3e wall us/step %.3f events %.3f frac %.4f traffic %s valu_busy %s" % (sys.argv[1].split("/")[-1], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"]["valu"] or {}).get("valu_busy_frac")))
if "env_api" in d: print("   env_api", {k: (round(v["us_per_step"], 2), v.get("ring")) for k, v in d["env_api"].items() if isinstance(v, dict)})
PY
done
done
