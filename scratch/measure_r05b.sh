#!/bin/bash
# Round 5, call b: the ring / RCCL / alias tests again (after the test fix), the load-imbalance probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ring.py tests/test_gpu_rccl.py tests/test_gpu_env_api.py -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest_ring.log 2>&1; echo "rc=$?" >> $O/pytest_ring.log
tail -25 $O/pytest_ring.log
timeout 300 python scratch/balance_probe.py 4096 > $O/balance_4096.txt 2>&1
cat $O/balance_4096.txt
timeout 300 python scratch/balance_probe.py 1024 > $O/balance_1024.txt 2>&1
cat $O/balance_1024.txt
