#!/bin/bash
for lib in cap3 cap4; do
 cp scratch/libcagpu_$lib.so gym_collision_avoidance_amd/libcagpu.so
 for epw in 2 3; do
 for mode in step rollout; do
  CAGPU_EPW=$epw timeout 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --mode $mode 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;d=json.loads(open('/tmp/b.json').read());print('$lib EPW=$epw', '$mode', round(d['ms_per_step']*1e3,2),'us/step')"
 done
 done
done
