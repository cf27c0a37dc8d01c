#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ring.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
