#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_env_api.py -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; tail -30 $O/pytest.log
timeout 300 python bench.py --steps 500 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','ranks_seen','per_rank_event_ms_per_step','stats_allreduce_us')}); print(d.get('env_api')); print(d['cpu_baseline'])"
