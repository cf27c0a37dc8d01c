#!/bin/bash
for nt in 192 256 320; do
  CAGPU_NT=$nt timeout 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;d=json.loads(open('/tmp/b.json').read());print('NT=$nt step', round(d['ms_per_step']*1e3,2),'us/step; rollout', round(d['rollout']['ms_per_step']*1e3,2))"
done
