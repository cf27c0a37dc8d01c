#!/bin/bash
# tile geometry of the N = 10 kernels (knobs build: CAGPU_TILE only selects the instantiation / launch geometry)
R=$PWD; O=$R/gpurun_out/tile; mkdir -p $O
KN=$R/gym_collision_avoidance_amd/libcagpu_knobs.so
for t in ${TILES:-0 6}; do for e in ${ENVS:-4096 8192 32768}; do
  CAGPU_LIB=$KN CAGPU_TILE=$t timeout 200 python bench.py --envs $e --steps 500 --warmup 50 --no-cpu-baseline > $O/t${t}_e$e.json 2> $O/t${t}_e$e.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/t${t}_e$e.json").read().strip().splitlines()[-1])
    print("tile $t envs $e: step %.2f us  rollout %.2f us/step  %s" % (d["event_ms_per_step"]*1e3, d.get("rollout",{}).get("ms_per_step",0)*1e3, d["roofline"]["kernel"][:70]))
except Exception as ex:
    print("tile $t envs $e FAILED", ex, open("$O/t${t}_e$e.err").read()[-300:])
PY
done; done
