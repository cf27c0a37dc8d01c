#!/bin/bash
# Round 6, call i: the integer item decomposition of P2b (kMagic20) + host-divided reciprocals in the pipelined kernels:
# bit-exactness tests on the experiment build, then same-box A/B against the product library.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06i
mkdir -p $O
cd $R
export TMPDIR=/tmp
G=$R/gym_collision_avoidance_amd
CAGPU_LIB=$G/libcagpu_fast.so timeout 1200 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity.py tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider --timeout 900 -k "(ring or pipelin or metric or bench_kernel or orca_velocities or 1024 or 4096x10 or 32768 or sharded) and not ga3c and not big and not n6 and not ragged and not config3 and not config5" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log | cut -c1-250
B="python bench.py --no-cpu-baseline --no-extras --min-timed-seconds 0.4"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-10s %-13s E %5d value %.3e wall us/step %.3f events us/step %.3f" % (sys.argv[2], d["config"]["launch_mode"], d["config"]["envs_per_gpu"], d["value"], d["ms_per_step"] * 1e3, d["event_ms_per_step"] * 1e3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do
for v in product fast; do
  L=$G/libcagpu_$v.so; [ "$v" = product ] && L=$G/libcagpu.so
  CAGPU_LIB=$L timeout 120 $B --steps 20 --warmup 5 > "$O/l20_${v}_$rep.json" 2> "$O/l20_${v}_$rep.err"; show "$O/l20_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 200 --lookahead 50 > "$O/l50_${v}_$rep.json" 2> "$O/l50_${v}_$rep.err"; show "$O/l50_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 2000 --mode rollout > "$O/ro_${v}_$rep.json" 2> "$O/ro_${v}_$rep.err"; show "$O/ro_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --steps 500 --mode step > "$O/st_${v}_$rep.json" 2> "$O/st_${v}_$rep.err"; show "$O/st_${v}_$rep.json" "$v"
  CAGPU_LIB=$L timeout 120 $B --envs 1024 --steps 640 > "$O/e1024_${v}_$rep.json" 2> "$O/e1024_${v}_$rep.err"; show "$O/e1024_${v}_$rep.json" "$v"
done
done
