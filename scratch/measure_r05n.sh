#!/bin/bash
# Round 5, the GA3C-CADRL re-measurement after the ga3c_kernel work (LSTM on bf16 MFMAs only, hand-placed pipelines): the
# network's tests, the same-box A/B against the kernel the round started with (libcagpu_r05start.so, built from commit
# df08dd3's csrc), the in-kernel phase / LSTM segment timers (ablate build), the counter passes, the config-3 bench line and
# its rocprofv3 trace -> gpurun_out/r05 (profiles/make_r05.py files them under profiles/r05_*).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -k "ga3c or checkpoint or config3 or episode" > $O/pytest_ga3c.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ga3c.log
tail -2 $O/pytest_ga3c.log
bash scratch/ga3c_ab3.sh gym_collision_avoidance_amd/libcagpu.so gym_collision_avoidance_amd/libcagpu_r05start.so > $O/ga3c_ab_final.txt 2>&1
cat $O/ga3c_ab_final.txt
CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_ablate.so timeout 300 python scratch/ga3c_phases.py > $O/ga3c_phases_final.txt 2>&1
WARM=150 CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_ablate.so timeout 300 python scratch/ga3c_phases.py > $O/ga3c_phases_final_steady.txt 2>&1
tail -18 $O/ga3c_phases_final_steady.txt
bash scratch/ga3c_pmc.sh final > $O/ga3c_pmc_final.txt 2>&1
CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_r05start.so bash scratch/ga3c_pmc.sh start > $O/ga3c_pmc_start.txt 2>&1
grep -h "SQ_BUSY_CYCLES\|MFMA_BUSY\|ACTIVE_INST_VALU\|INSTS_VALU \|ga3c_kernel" $O/ga3c_pmc_final.txt $O/ga3c_pmc_start.txt
timeout 600 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_ga3c20.json 2> $O/cfg3.err
CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_r05start.so timeout 600 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline > $O/cfg3_ga3c20_r05start_samebox.json 2> $O/cfg3_start.err
cut -c1-260 $O/cfg3_ga3c20.json; cut -c1-260 $O/cfg3_ga3c20_r05start_samebox.json
cd /tmp
rm -rf $O/prof_ga3c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ga3c -- python $R/bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline --min-timed-seconds 0 > $O/prof_ga3c.log 2>&1
find $O -name '*agent_info.csv' -delete
find $O -name '*kernel_trace.csv' -size +8M -delete
find $R/gpurun_out/ga3c_pmc_final $R/gpurun_out/ga3c_pmc_start -name '*kernel_trace.csv' -delete
