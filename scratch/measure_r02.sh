#!/bin/bash
# One gpurun call: GPU test-suite, the driver-shaped bench line, the default bench line, then the profiler passes
# (each from /tmp with TMPDIR=/tmp; counters in their own --pmc passes, never combined with traces).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
cat $O/bench_driver.json | cut -c1-600
cat $O/bench_n1.json | cut -c1-1500
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $B > $O/prof_stats.log 2>&1
P="$B --steps 300 --warmup 500 --min-warm-seconds 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $P > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $P > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prof_sq -- $P > $O/prof_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/prof_sq2 -- $P > $O/prof_sq2.log 2>&1
# csv trees are large: keep only what profiles/summarize.py reads
find $O -name '*agent_info.csv' -delete
ls -la $O
du -sh $O
# the "next"-row workloads (their own bench lines; profiles/r02_bench_*.json)
cd $R
timeout 600 python bench.py --workload ga3c20 --steps 100 --warmup 10 > $O/bench_ga3c20.json 2> $O/bench_ga3c20.err
timeout 600 python bench.py --workload crowd50_laser --steps 50 --warmup 5 > $O/bench_crowd50.json 2> $O/bench_crowd50.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ga3c -- python $R/bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_ga3c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_crowd -- python $R/bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_crowd.log 2>&1
find $O -name '*agent_info.csv' -delete
du -sh $O
