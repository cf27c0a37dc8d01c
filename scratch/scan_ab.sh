#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 -k "laser" 2>&1 | tail -4
for rep in 1 2; do
for lib in gym_collision_avoidance_amd/libcagpu_base2.so gym_collision_avoidance_amd/libcagpu.so; do
  CAGPU_LIB=$lib timeout 300 python bench.py --workload crowd50_laser --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-50s step %.1f us scan %.1f us  total %.3f ms  frac %.4f' % ('$lib', r['step_kernel_us'], r['scan_kernel_us'], d['ms_per_step'], r['frac']))"
done; done
