#!/bin/bash
# scratch/ab_sweep.sh -- bench the experiment build (libcagpu_ablate_fast.so) under a list of CAGPU_ABLATE bit masks
# usage: ABS="0 8192 16384" TAG=prio bash scratch/ab_sweep.sh
R=$PWD
O=$R/gpurun_out/${TAG:-ab}
mkdir -p $O
AB=$R/gym_collision_avoidance_amd/libcagpu_${LIBV:-ablate_fast}.so
for ab in $ABS; do
  CAGPU_LIB=$AB CAGPU_ABLATE=$ab timeout 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/ab_$ab.json 2> $O/ab_$ab.err
done
ABS="$ABS" python - <<PY
import json, os
base = None
for ab in os.environ["ABS"].split():
    try:
        d = json.loads(open("$O/ab_%s.json" % ab).read().strip().splitlines()[-1])
        st, ro, ts = d["event_ms_per_step"] * 1e3, d["rollout"]["ms_per_step"] * 1e3, d["two_streams"]["ms_per_step"] * 1e3
        if base is None: base = (st, ro)
        print("ablate %-8s step %.2f us (%+.2f)  rollout %.2f us/step (%+.2f)  two-streams %.2f" % (ab, st, st - base[0], ro, ro - base[1], ts))
    except Exception as e:
        print(ab, "FAILED", e, open("$O/ab_%s.err" % ab).read()[-400:])
PY
