#!/bin/bash
# scratch/lib_sweep.sh -- bench a list of experiment builds of the library (compile-time switches, product-like code)
# usage: LIBS="exp0_fast exp1_fast" TAG=prio bash scratch/lib_sweep.sh      ("product" = libcagpu.so)
R=$PWD
O=$R/gpurun_out/${TAG:-libs}
mkdir -p $O
for v in $LIBS; do
  L=$R/gym_collision_avoidance_amd/libcagpu_$v.so
  [ "$v" = product ] && L=$R/gym_collision_avoidance_amd/libcagpu.so
  CAGPU_LIB=$L timeout 200 python bench.py --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline ${BARGS} > $O/$v.json 2> $O/$v.err
done
LIBS="$LIBS" python - <<PY
import json, os
base = None
for v in os.environ["LIBS"].split():
    try:
        d = json.loads(open("$O/%s.json" % v).read().strip().splitlines()[-1])
        st, ro, ts = d["event_ms_per_step"] * 1e3, d.get("rollout", {}).get("ms_per_step", 0) * 1e3, d.get("two_streams", {}).get("ms_per_step", 0) * 1e3
        if base is None: base = (st, ro)
        print("%-16s step %.2f us (%+.2f)  rollout %.2f us/step (%+.2f)  two-streams %.2f" % (v, st, st - base[0], ro, ro - base[1], ts))
    except Exception as e:
        print(v, "FAILED", e, open("$O/%s.err" % v).read()[-400:])
PY
