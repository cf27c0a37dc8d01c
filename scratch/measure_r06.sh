#!/bin/bash
# Round 6, the measurement call (like scratch/measure_r05.sh): GPU test-suite, the driver-shaped and default bench lines + one
# launch per step + fused rollout, the 2-rank and one-rank-RCCL lines, the profiler passes (each from /tmp with TMPDIR=/tmp;
# counters in their own --pmc passes, never combined with traces) -- the counter passes over the n-step kernel at the launch
# length the driver's command times (20 steps per launch) --, the counter calibration launches, one bench line per BASELINE
# config (config 3 also with fused sensing), the same-box A/B of this round's progress-fair priorities, and the kernel traces of
# the config-3 / config-5 runs.  Every bench line carries `provenance` (library sha256, git commit of the build);
# profiles/make_r06.py refuses to file lines of different libraries together -- and profiler output of more than one call: gpurun
# MERGES a call's files into the local gpurun_out/, so `rm -rf gpurun_out/r06` in the build container before this call.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
rm -rf $O
mkdir -p $O
cd $R
export TMPDIR=/tmp
python - > $O/provenance.json <<'PY'
import json, bench
print(json.dumps(bench.provenance()))
PY
cat $O/provenance.json
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --share-device --backend gloo --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-extras > $O/bench_force_dist_rccl.json 2> $O/bench_force_dist.err
timeout 300 python bench.py --mode step --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
timeout 300 python bench.py --mode rollout --no-cpu-baseline --no-extras > $O/bench_rollout.json 2> $O/bench_rollout.err
timeout 300 python bench.py --steps 2000 --lookahead 200 --no-cpu-baseline --no-extras > $O/bench_lookahead200.json 2> $O/bench_lookahead200.err
cut -c1-300 $O/bench_driver.json
# ---- same-box A/B: this round's progress-fair priorities off (the same source, -DCAGPU_PIPE_YIELD_T=0) and the round-5 end state
G=$R/gym_collision_avoidance_amd
BA="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
for v in product "dPIPE_YIELD_T=0" r05end_fast; do
  L=$G/libcagpu_$v.so; [ "$v" = product ] && L=$G/libcagpu.so
  [ -f "$L" ] || continue
  CAGPU_LIB=$L timeout 120 $BA --steps 20 --warmup 5 > "$O/ab_l20_$v.json" 2> "$O/ab_l20_$v.err"
  CAGPU_LIB=$L timeout 120 $BA --steps 200 --lookahead 50 > "$O/ab_l50_$v.json" 2> "$O/ab_l50_$v.err"
  CAGPU_LIB=$L timeout 120 $BA --steps 2000 --mode rollout > "$O/ab_ro_$v.json" 2> "$O/ab_ro_$v.err"
done
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $B > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_driver -- $B --steps 20 --warmup 5 --min-timed-seconds 0.2 > $O/prof_stats_driver.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_step -- $B --mode step > $O/prof_stats_step.log 2>&1
# counters over launches of 20 steps (the length the driver's command times): 25 timed launches behind 25 warm-up launches
P="$B --steps 500 --lookahead 20 --warmup 500 --min-warm-seconds 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $P > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $P > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prof_sq -- $P > $O/prof_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/prof_sq2 -- $P > $O/prof_sq2.log 2>&1
S="$B --mode step --steps 300 --warmup 500 --min-warm-seconds 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch_step -- $S > $O/prof_fetch_step.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write_step -- $S > $O/prof_write_step.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq_step -- $S > $O/prof_sq_step.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- python $R/scratch/copy8_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- python $R/scratch/copy8_calib.py > $O/calib_write.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R
timeout 300 python bench.py --envs 1024 --steps 640 --no-cpu-baseline > $O/cfg2_1024x10.json 2> $O/cfg2.err
timeout 300 python bench.py --envs 1024 --mode step --no-cpu-baseline --no-extras > $O/cfg2_1024x10_step.json 2> $O/cfg2_step.err
timeout 600 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_ga3c20.json 2> $O/cfg3.err
timeout 600 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline --ga3c-fused > $O/cfg3_ga3c20_fused.json 2> $O/cfg3_fused.err
timeout 300 python bench.py --envs 32768 --steps 200 --lookahead 50 --warmup 50 --no-cpu-baseline --no-extras > $O/cfg4_32768x10_one_gpu.json 2> $O/cfg4.err
timeout 300 python bench.py --envs 32768 --mode step --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/cfg4_32768x10_one_gpu_step.json 2> $O/cfg4_step.err
timeout 600 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline > $O/cfg5_crowd50.json 2> $O/cfg5.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ga3c -- python $R/bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline --min-timed-seconds 0 > $O/prof_ga3c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_crowd -- python $R/bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline --min-timed-seconds 0 > $O/prof_crowd.log 2>&1
find $O -name '*agent_info.csv' -delete
find $O -name '*kernel_trace.csv' -size +8M -delete
cd $R
python profiles/summarize.py $O/prof_stats $O/prof_stats_driver $O/prof_stats_step $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_sq2 $O/prof_fetch_step $O/prof_write_step $O/prof_sq_step $O/prof_ga3c $O/prof_crowd > $O/summary.md 2>&1
du -sh $O
