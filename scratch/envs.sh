#!/bin/bash
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
for e in 3072 4096 4608 5120 6144 8192 32768; do
  timeout 200 python bench.py --envs $e --steps 500 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;d=json.loads(open('/tmp/b.json').read());print('E=$e step', round(d['ms_per_step']*1e3,2),'us/step', round(d['value']/1e6,1), 'M agent-steps/s; rollout', round(d['rollout']['ms_per_step']*1e3,2))"
done
