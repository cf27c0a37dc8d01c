#!/bin/bash
# Round 4, the measurement call: GPU test-suite (+ a soak of the pipeline parity tests), the driver-shaped and default bench
# lines, the 2-rank self-spawned line, the profiler passes (each from /tmp with TMPDIR=/tmp; counters in their own --pmc
# passes, never combined with traces), one bench line per BASELINE config, the GA3C rows-vs-time record.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider -k "pipelined or orca_velocities or metric or config" >> $O/soak.log 2>&1; done
grep -E "passed|failed" $O/soak.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --share-device --backend gloo --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
timeout 600 python bench.py --no-pipeline --no-cpu-baseline > $O/bench_n1_nopipe.json 2> $O/bench_n1_nopipe.err
cut -c1-300 $O/bench_driver.json
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $B > $O/prof_stats.log 2>&1
P="$B --steps 300 --warmup 500 --min-warm-seconds 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $P > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $P > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prof_sq -- $P > $O/prof_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/prof_sq2 -- $P > $O/prof_sq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq_rollout -- $P --mode rollout > $O/prof_sq_rollout.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R
timeout 300 python bench.py --envs 1024 --no-cpu-baseline --no-extras > $O/cfg2_1024x10.json 2> $O/cfg2.err
timeout 600 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_ga3c20.json 2> $O/cfg3.err
timeout 300 python bench.py --envs 32768 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/cfg4_32768x10_one_gpu.json 2> $O/cfg4.err
timeout 600 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_crowd50.json 2> $O/cfg5.err
timeout 300 python bench.py --mode rollout --no-cpu-baseline --no-extras > $O/bench_rollout.json 2> $O/bench_rollout.err
timeout 300 python scratch/ga3c_rows.py > $O/ga3c_rows.json 2> $O/ga3c_rows.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ga3c -- python $R/bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline --min-timed-seconds 0 > $O/prof_ga3c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_crowd -- python $R/bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline --min-timed-seconds 0 > $O/prof_crowd.log 2>&1
find $O -name '*agent_info.csv' -delete
# keep the merged output small: the kernel-trace csv of the long runs is not needed (the stats csv is)
find $O -name '*kernel_trace.csv' -size +8M -delete
cd $R
bash scratch/ga3c_pmc.sh > $O/ga3c_pmc.txt 2>&1
cd $R
python profiles/summarize.py $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_sq2 $O/prof_sq_rollout $O/prof_ga3c $O/prof_crowd > $O/summary.md 2>&1
du -sh $O
