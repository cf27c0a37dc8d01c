#!/bin/bash
# Round 6, call e: GPU suite with the row-packing fold (CaOut.ga3c_rows), the side-stream fault probe and prepared ring launches;
# config 3 against the build before the fold (same box); the driver-shaped line + its kernel trace (host overhead of a block).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log | cut -c1-300
G=$R/gym_collision_avoidance_amd
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print("%-26s %-13s value %.3e wall us/step %.3f events %.3f | %s avg launch %.1f us %s" % (sys.argv[2], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3, r["kernel"][:28], r["avg_launch_us"], ("rows %d" % r["rows_evaluated"]) if "rows_evaluated" in r else ""))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for v in product "dPIPE_YIELD_T=0"; do
  L=$G/libcagpu_$v.so; [ "$v" = product ] && L=$G/libcagpu.so
  CAGPU_LIB=$L timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > "$O/cfg3_${v}_$rep.json" 2> "$O/cfg3_${v}_$rep.err"; show "$O/cfg3_${v}_$rep.json" "cfg3 $v"
done
done
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/driver_$rep.json 2> $O/driver_$rep.err; show $O/driver_$rep.json "driver shape"
done
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/n1.json 2> $O/n1.err; show $O/n1.json "default"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --min-timed-seconds 0.2 > $O/prof_driver.log 2>&1
cd $R
python profiles/summarize.py $O/prof_driver | head -12
