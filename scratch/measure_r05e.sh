#!/bin/bash
# Round 5, call e: the whole GPU suite on the new tree; ga3c_kernel phase timers + staggered co-resident workgroups A/B;
# configs 2 / 4 through the look-ahead ring; the counter calibration launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_ablate.so timeout 300 python scratch/ga3c_phases.py > $O/ga3c_phases.txt 2>&1
cat $O/ga3c_phases.txt
for rep in 1 2; do
  for lib in libcagpu libcagpu_dGA3C_STAGGER=4 libcagpu_dGA3C_STAGGER=7 libcagpu_dGA3C_STAGGER=10; do
    CAGPU_LIB=$R/gym_collision_avoidance_amd/$lib.so timeout 300 python scratch/ga3c_rows.py > $O/rows_${lib}_$rep.json 2> $O/rows_${lib}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05e/rows_*.json")):
    try:
        d = json.load(open(f))
        print("%-44s" % f.split("/")[-1], "mean %.1f us;" % d["us_mean"], " ".join("%d:%.0f" % (r["rows"], r["us"]) for r in d["by_rows"][::2]))
    except Exception as e:
        print(f, "failed", e)
PY
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.3"
for M in "--envs 32768 --steps 100 --lookahead 20 --warmup 20" "--envs 32768 --steps 100 --lookahead 50 --warmup 50" "--envs 32768 --mode step --steps 100 --warmup 20" "--envs 32768 --mode rollout --steps 300 --warmup 20" "--envs 1024 --steps 640" "--envs 1024 --mode step --steps 500"; do
  timeout 200 $B $M > $O/x.json 2> $O/x.err
  python - "$O/x.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("E %5d %-13s value %.3e wall us/step %.3f events us/step %.3f frac %.4f  %s" % (d["config"]["envs_per_gpu"], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"]["kernel"][:60]))
PY
done
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- python $R/scratch/copy8_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- python $R/scratch/copy8_calib.py > $O/calib_write.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("calib_fetch", "calib_write"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r05e/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(d, k, "n=%d" % len(v), "mean %.1f min %.1f max %.1f" % (sum(v) / len(v), min(v), max(v)))
PY
du -sh $O
