#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 120 ./scratch/floor2 > $O/floor2.txt 2>&1; cat $O/floor2.txt
CAGPU_LIB=gym_collision_avoidance_amd/libcagpu_pipetime_fast.so timeout 300 python scratch/pipetime.py > $O/pipetime.txt 2>&1; cat $O/pipetime.txt
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "library_loads or without_precomputed or ga3c_graph or hip_network" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
