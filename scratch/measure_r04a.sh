#!/bin/bash
# round 4, call A: the new tests (bench geometries, swap closure, lean range, self-spawning bench), the whole GPU suite,
# the driver-shaped + default bench lines, and the small-tile A/B (product vs -DCAGPU_PIPE_TE_MIN=4) at 1024 / 2048 / 3072 envs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider --timeout 600 -x --deselect tests/test_gpu_bench_geometry.py::test_swap_cases_agree_on_the_gpus_libm_bits > $O/geom.log 2>&1
echo "geom rc=$?" >> $O/geom.log
timeout 600 python -m pytest tests/test_gpu_bench_geometry.py -m gpu -q -p no:cacheprovider --timeout 500 -k swap_cases > $O/swap.log 2>&1
echo "swap rc=$?" >> $O/swap.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "lean_divide" > $O/lean.log 2>&1
echo "lean rc=$?" >> $O/lean.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --deselect tests/test_gpu_bench_geometry.py > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
for E in 1024 2048 3072; do
  for rep in 1 2; do
    timeout 200 python bench.py --no-cpu-baseline --no-extras --envs $E > $O/te_auto_${E}_$rep.json 2>/dev/null
    CAGPU_LIB=$PWD/gym_collision_avoidance_amd/libcagpu_dPIPE_TE_MIN=4.so timeout 200 python bench.py --no-cpu-baseline --no-extras --envs $E > $O/te_four_${E}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04a/*.json")):
    try:
        d=json.load(open(f)); print("%-34s step %.2f us (events %.2f) %s  rollout %.2f" % (f.split("/")[-1], d["ms_per_step"]*1e3, d["event_ms_per_step"]*1e3, d["roofline"]["kernel"][:34], d.get("rollout",{}).get("ms_per_step",0)*1e3))
    except Exception as e: print(f, "failed", e)
PY
tail -3 $O/geom.log $O/swap.log $O/lean.log $O/pytest_gpu.log
