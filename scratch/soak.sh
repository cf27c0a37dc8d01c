#!/bin/bash
# the pipeline parity tests several times over (the arrival-counter hand-overs must hold on every launch, not on most)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/soak.log
for i in $(seq 1 ${1:-6}); do
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 500 -k "pipelin or orca_velocities or metric_geometry or rollout or plan or ragged or rewrites" >> gpurun_out/soak.log 2>&1
  echo "run $i rc=$?" >> gpurun_out/soak.log
done
grep "rc=\|passed\|failed" gpurun_out/soak.log
