#!/bin/bash
# Round 5, call a: the look-ahead ring (tests + first bench lines + ring-length sweep), RCCL once, the VALU issue-cost
# micro-benchmark, and the in-situ busy-cycle cost of the n-step kernel's phases (dPIPE_DUP builds, SQ_ACTIVE_INST_VALU).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_rccl.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/pytest_ring.log 2>&1; echo "rc=$?" >> $O/pytest_ring.log
tail -15 $O/pytest_ring.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
cut -c1-400 $O/bench_driver.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
cut -c1-400 $O/bench_n1.json
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.3"
for L in 5 10 20 32 50 100 200; do
  for E in 4096 1024; do
    timeout 120 $B --envs $E --steps $((L * 10)) --lookahead $L > $O/sweep_${E}_L$L.json 2> $O/sweep_${E}_L$L.err
    python - "$O/sweep_${E}_L$L.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("E %5d %-13s value %.3e wall us/step %.3f events us/step %.3f frac %.4f" % (d["config"]["envs_per_gpu"], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3, d["roofline"]["frac"]))
PY
  done
done
timeout 120 $B --mode step > $O/bench_step.json 2> $O/bench_step.err
timeout 120 $B --mode step --envs 1024 > $O/bench_step_1024.json 2> $O/bench_step_1024.err
timeout 120 $B --mode rollout > $O/bench_rollout.json 2> $O/bench_rollout.err
timeout 120 $B --mode rollout --envs 1024 > $O/bench_rollout_1024.json 2> $O/bench_rollout_1024.err
for f in bench_step bench_step_1024 bench_rollout bench_rollout_1024; do python - $O/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-40s E %5d %-10s value %.3e us/step %.3f events %.3f" % (sys.argv[1].split("/")[-1], d["config"]["envs_per_gpu"], d["config"]["launch_mode"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
PY
done
# ---- VALU issue costs
timeout 120 scratch/valu_rates > $O/valu_rates.txt 2>&1
cat $O/valu_rates.txt
# ---- in-situ busy cycles of the n-step kernel's phases
cd /tmp
for lib in fast dPIPE_DUP=1,fast dPIPE_DUP=2,fast dPIPE_DUP=4,fast dPIPE_DUP=8,fast dPIPE_DUP=16,fast dPIPE_DUP=32,fast; do
  P="python $R/bench.py --no-cpu-baseline --no-extras --steps 500 --lookahead 50 --warmup 500 --min-warm-seconds 0 --min-timed-seconds 0"
  CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_$lib.so timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d "$O/dup/$lib" -- $P > "$O/dup_$lib.log" 2>&1
done
find $O -name '*agent_info.csv' -delete
cd $R
python - <<'PY' | tee $O/dup_busy.txt
import csv, glob, os, collections
base = None
rows = []
for d in sorted(glob.glob("gpurun_out/r05a/dup/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ca_pipe_kernel<10, 4, true>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v[2:]) / max(1, len(v[2:])) / 50.0 for k, v in acc.items()}   # per STEP (50 steps per launch)
    rows.append((os.path.basename(d.rstrip("/")), m, len(acc.get("SQ_INSTS_VALU", []))))
base = [m for n, m, _ in rows if n == "fast"][0]
print("# per step of ca_pipe_kernel<10,4,true> (4096 x 10, 50 steps per launch); busy = SQ_ACTIVE_INST_VALU x 4 / 1024 = cycles per SIMD")
for n, m, cnt in rows:
    busy, b0 = m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024, base.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024
    print("%-22s VALU %9.0f (%+8.0f) busy cycles/SIMD %7.0f (%+6.0f) cyc/inst of the delta %5.2f  SALU %9.0f LDS %8.0f wave-cycles %10.0f wait_any %10.0f (n=%d)" % (
        n, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_VALU", 0) - base.get("SQ_INSTS_VALU", 0), busy, busy - b0,
        (busy - b0) * 1024 / max(1.0, m.get("SQ_INSTS_VALU", 0) - base.get("SQ_INSTS_VALU", 0)), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0),
        m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_ANY", 0), cnt))
PY
du -sh $O
