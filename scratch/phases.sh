#!/bin/bash
# scratch/phases.sh -- in-kernel phase timers (ablate build) for the single-step and the rollout kernel
R=$PWD
O=$R/gpurun_out/${TAG:-phases}
mkdir -p $O
AB=$R/gym_collision_avoidance_amd/libcagpu_ablate_fast.so
CAGPU_LIB=$AB timeout 300 python scratch/prof_phases.py > $O/phases_step.txt 2>&1
CAGPU_LIB=$AB MODE=rollout timeout 300 python scratch/prof_phases.py > $O/phases_rollout.txt 2>&1
cat $O/phases_step.txt; cat $O/phases_rollout.txt
