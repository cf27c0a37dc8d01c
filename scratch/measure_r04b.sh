#!/bin/bash
# round 4, call B: the bench-geometry tests (configs 2 - 5, shards, swap closure) on the library without the small tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider --timeout 900 > $O/geom.log 2>&1
echo "geom rc=$?" >> $O/geom.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_api.py -m gpu -q -p no:cacheprovider -k "lean_divide or pipelined or orca_velocities or bench" > $O/sel.log 2>&1
echo "sel rc=$?" >> $O/sel.log
tail -n 25 $O/geom.log
tail -n 5 $O/sel.log
