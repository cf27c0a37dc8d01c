#!/bin/bash
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
for e in 2048 4096 6144 8192 32768; do for cfg in "" "CAGPU_TILE=4" "CAGPU_TILE=4 CAGPU_NOSTAGE=1"; do
  r=$(env $cfg timeout 300 python bench.py --envs $e --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['rollout']['ms_per_step']*1e3,2))")
  echo "E=$e [$cfg] step / rollout us: $r"
done; done
