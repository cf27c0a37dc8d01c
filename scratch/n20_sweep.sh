#!/bin/bash
# config 3's step kernel (N = 20, no ORCA): tile size / threads per workgroup (knobs build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
KN=gym_collision_avoidance_amd/libcagpu_knobs.so
for t in 3 2 1; do for nt in 256 512; do
  CAGPU_LIB=$KN CAGPU_TILE=$t CAGPU_NT=$nt timeout 300 python bench.py --workload ga3c20 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $t nt $nt: %.1f us / step' % (d['event_ms_per_step']*1e3))"
done; done
