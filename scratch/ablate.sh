#!/bin/bash
# builds nothing; expects gym_collision_avoidance_amd/libcagpu.so compiled with -DCAGPU_ABLATE
for ab in 0 1 2 4 8 32 64 128; do
  for mode in step rollout; do
  CAGPU_ABLATE=$ab timeout 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --mode $mode 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;d=json.loads(open('/tmp/b.json').read());print('ablate=$ab', '$mode', round(d['ms_per_step']*1e3,2),'us/step')"
  done
done
