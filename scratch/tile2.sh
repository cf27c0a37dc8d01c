#!/bin/bash
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
for cfg in "" "CAGPU_NOSTAGE=1" "CAGPU_NOSTAGE=1 CAGPU_TILE=5" "CAGPU_NOSTAGE=1 CAGPU_TILE=4" "CAGPU_NOSTAGE=1 CAGPU_TILE=3" "CAGPU_NOSTAGE=1 CAGPU_TILE=2" "CAGPU_TILE=4"; do
  r=$(env $cfg timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['rollout']['ms_per_step']*1e3,2), d['episode_stats']['episodes'])")
  echo "[$cfg] step / rollout us, episodes: $r"
done
