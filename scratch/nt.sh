#!/bin/bash
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2>&1   # warm the box
for nt in default 128 256; do
  if [ $nt = default ]; then unset CAGPU_NT; else export CAGPU_NT=$nt; fi
  timeout 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;d=json.loads(open('/tmp/b.json').read());print('NT=$nt step', round(d['ms_per_step']*1e3,2),'us/step; rollout', round(d['rollout']['ms_per_step']*1e3,2))"
done
