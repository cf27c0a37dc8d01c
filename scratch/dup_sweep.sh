#!/bin/bash
# scratch/dup_sweep.sh -- cost of each pair phase in situ: run it twice (ablate build, CAGPU_ABLATE=bit) and compare
R=$PWD
O=$R/gpurun_out/${TAG:-dup}
mkdir -p $O
AB=$R/gym_collision_avoidance_amd/libcagpu_ablate_fast.so
for ab in 0 256 512 2048 4096 2 32 64 1; do
  CAGPU_LIB=$AB CAGPU_ABLATE=$ab timeout 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/ab_$ab.json 2> $O/ab_$ab.err
done
python - <<PY
import json
base = None
names = {0: "baseline", 256: "P2 twice", 512: "P2b twice", 2048: "P3 twice", 4096: "P4 twice", 2: "no LP3", 32: "no P3", 64: "no P4", 1: "no ORCA"}
for ab in (0, 256, 512, 2048, 4096, 2, 32, 64, 1):
    try:
        d = json.loads(open("$O/ab_%d.json" % ab).read().strip().splitlines()[-1])
        st, ro = d["event_ms_per_step"] * 1e3, d["rollout"]["ms_per_step"] * 1e3
        if base is None: base = (st, ro)
        print("%-12s step %.2f us (%+.2f)  rollout %.2f us/step (%+.2f)" % (names[ab], st, st - base[0], ro, ro - base[1]))
    except Exception as e:
        print(ab, "FAILED", e)
PY
