#!/bin/bash
# tile-size x workgroup-size sweep of the default step kernel (4096 x 10)
for t in 6 5 4 3 2; do for nt in 128 256; do
  r=$(CAGPU_TILE=$t CAGPU_NT=$nt timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['rollout']['ms_per_step']*1e3,2))")
  echo "tile=$t nt=$nt step/rollout us: $r"
done; done
