#!/bin/bash
# scratch/measure_all.sh -- one gpurun call: GPU test-suite, the three bench workloads, rocprofv3 summaries.
# usage (from the repo root on the GPU box): bash scratch/measure_all.sh
R=$PWD
O=$R/gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/test_gpu.log 2>&1
tail -3 $O/test_gpu.log
timeout 600 python bench.py > $O/bench_rvo10.json 2> $O/bench_rvo10.err
timeout 600 python bench.py --workload ga3c20 --steps 300 --warmup 30 > $O/bench_ga3c20.json 2> $O/bench_ga3c20.err
timeout 600 python bench.py --workload crowd50_laser --steps 100 --warmup 10 > $O/bench_crowd50.json 2> $O/bench_crowd50.err
python - <<'PY'
import json
for n in ("rvo10", "ga3c20", "crowd50"):
    try:
        d = json.loads(open("gpurun_out/bench_%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, "%.3e" % d["value"], d["unit"], "ms/step %.4f" % d["ms_per_step"], r["bound"], "%.3f" % r["achieved"], r["unit"],
              "frac %.4f" % r["frac"], "launch us %.1f" % r["avg_launch_us"], {k: round(v, 1) for k, v in r.items() if k.endswith("_kernel_us")})
    except Exception as e:
        print(n, "FAILED", e, open("gpurun_out/bench_%s.err" % n).read()[-800:])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python $R/bench.py --steps 300 --warmup 1500 --no-cpu-baseline --no-extras > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python $R/bench.py --steps 300 --warmup 1500 --no-cpu-baseline --no-extras > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq -- python $R/bench.py --steps 300 --warmup 1500 --no-cpu-baseline --no-extras > $O/prof_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ga3c -- python $R/bench.py --workload ga3c20 --steps 100 --warmup 10 > $O/prof_ga3c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_crowd -- python $R/bench.py --workload crowd50_laser --steps 50 --warmup 5 > $O/prof_crowd.log 2>&1
cd $R
# keep only the small csv summaries (kernel_stats, counter_collection is large -> summarised here)
python profiles/summarize.py $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq > $O/summary_rvo10.md 2>&1
python profiles/summarize.py $O/prof_ga3c > $O/summary_ga3c20.md 2>&1
python profiles/summarize.py $O/prof_crowd > $O/summary_crowd50.md 2>&1
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
head -12 $O/summary_ga3c20.md; head -12 $O/summary_crowd50.md
