#!/bin/bash
export CAGPU_G16=1
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  " | tail -40) > gpurun_out/test_gpu.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/test_gpu.log | tail -5
grep -E "^E  " gpurun_out/test_gpu.log | head -20
for epw in 1 2; do
  CAGPU_EPW=$epw timeout 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;d=json.loads(open('/tmp/b.json').read());print('G16 EPW=$epw step', round(d['ms_per_step']*1e3,2),'us/step; rollout', round(d['rollout']['ms_per_step']*1e3,2))"
done
