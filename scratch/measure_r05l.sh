#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 900 -k "big_envs or too_many or bad_arguments" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
