#!/bin/bash
# the network with all NINE plane products (libcagpu_nine.so: every float32 product exact) against the product's six: launch
# time by rows and the error against a float64 evaluation (test_ga3c_split_operand_network_is_float32_accurate prints it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/ga3c_nine
rm -rf $O; mkdir -p $O
NINE=$PWD/gym_collision_avoidance_amd/libcagpu_nine.so
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -s -k "split_operand" 2>&1 | grep "logits up to"
CAGPU_LIB=$NINE timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -s -k "split_operand or ga3c_logits" 2>&1 | grep "logits up to\|passed\|failed"
for rep in 1 2; do
  timeout 300 python scratch/ga3c_rows.py > $O/rows_six_$rep.json 2> $O/rows_six_$rep.err
  CAGPU_LIB=$NINE timeout 300 python scratch/ga3c_rows.py > $O/rows_nine_$rep.json 2> $O/rows_nine_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ga3c_nine/rows_*.json")):
    d = json.load(open(f))
    print("%-22s" % f.split("/")[-1], "mean %.1f us;" % d["us_mean"], " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"][::2]))
PY
