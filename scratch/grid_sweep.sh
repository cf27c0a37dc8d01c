#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for E in 4096 6144 8192 16384 32768; do
for lib in gym_collision_avoidance_amd/libcagpu.so "gym_collision_avoidance_amd/libcagpu_dPIPE_ANYGRID=1,fast.so"; do
for mode in step rollout; do
  CAGPU_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-extras --envs $E --steps 400 --warmup 50 --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('E=%6d %-8s %-55s %8.2f us/step  %.3e agent-steps/s  frac %.4f' % ($E, '$mode', d['roofline']['kernel'][:55], d['event_ms_per_step']*1e3, d['value'], d['roofline']['frac']))"
done; done; done
