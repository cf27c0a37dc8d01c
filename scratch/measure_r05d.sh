#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
export TMPDIR=/tmp
CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_steptime_fast.so timeout 300 python scratch/steptime.py 20 > $O/steptime_L20.txt 2>&1
cat $O/steptime_L20.txt
BALANCE=1 CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_steptime_fast.so timeout 300 python scratch/steptime.py 20 > $O/steptime_L20_bal1.txt 2>&1
cat $O/steptime_L20_bal1.txt
