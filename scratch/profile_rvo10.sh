#!/bin/bash
# only the rocprofv3 passes of the metric workload (see measure_all.sh for everything)
R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python $R/bench.py --steps 300 --warmup 1500 --no-cpu-baseline --no-extras > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python $R/bench.py --steps 300 --warmup 1500 --no-cpu-baseline --no-extras > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq -- python $R/bench.py --steps 300 --warmup 1500 --no-cpu-baseline --no-extras > $O/prof_sq.log 2>&1
cd $R
python profiles/summarize.py $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq > $O/summary_rvo10.md 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
grep "ca_kernel<256, true, 10, false, true" $O/summary_rvo10.md
python bench.py > $O/bench_rvo10.json 2> $O/bench_rvo10.err; tail -c 600 $O/bench_rvo10.json
