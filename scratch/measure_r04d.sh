#!/bin/bash
# round 4, call D: the new plugin-surface tests (per-agent GA3C checkpoints, batched stochastic RVO, host policies of a
# batch) + the whole GPU suite on the rebuilt library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -k "stochastic or checkpoints or heading_noise or host_path" > $O/new.log 2>&1
echo "new rc=$?" >> $O/new.log
tail -n 40 $O/new.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 15 $O/pytest_gpu.log
