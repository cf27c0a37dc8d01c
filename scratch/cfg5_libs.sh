#!/bin/bash
# config 5 (bench.py --workload crowd50_laser): the product against every libcagpu_*.so named on the command line, same box;
# the laser-scan parity tests run on each variant first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/cfg5_libs
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "laser or config5" > $O/tests_product.log 2>&1
echo "product tests rc=$? $(tail -n 1 $O/tests_product.log)"
for lib in "$@"; do
  n=$(basename $lib .so)
  CAGPU_LIB=$PWD/$lib timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "laser or config5" > $O/tests_$n.log 2>&1
  echo "$n tests rc=$? $(tail -n 1 $O/tests_$n.log)"
done
for rep in 1 2; do
  timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_product_$rep.json 2> $O/cfg5_product_$rep.err
  for lib in "$@"; do
    n=$(basename $lib .so)
    CAGPU_LIB=$PWD/$lib timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_${n}_$rep.json 2> $O/cfg5_${n}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg5_libs/cfg5_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print("%-30s" % f.split("/")[-1], "ms_per_step %.4f value %.3e" % (d["ms_per_step"], d["value"]), "step %.1f scan %.1f" % (r.get("step_kernel_us", 0), r.get("scan_kernel_us", 0)))
    except Exception as e:
        print(f, "failed", e)
PY
