#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
CAGPU_LIB=$R/gym_collision_avoidance_amd/libcagpu_pipetime_fast.so timeout 300 python scratch/pipetime_multi.py 20 > $O/pipetime_multi_L20.txt 2>&1
cat $O/pipetime_multi_L20.txt
