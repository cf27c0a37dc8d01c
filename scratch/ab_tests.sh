#!/bin/bash
# A/B of fast builds in one call + the pipeline parity tests on the product library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/sweep
bash scratch/lib_sweep2.sh "$@" > gpurun_out/ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "pipelin or orca_velocities or metric_geometry or rollout or plan or ragged or fuzz" >> gpurun_out/ab.log 2>&1
echo "pytest rc=$?" >> gpurun_out/ab.log
tail -30 gpurun_out/ab.log
