#!/bin/bash
# round 3, first GPU call: the whole GPU test-suite (new: pipelined kernel, orca_vel bit-exactness, ragged batches, the
# reference suite), then A/B timings of the step kernel variants in ONE call (boxes differ by up to 1 us)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -40 $O/pytest.log
for v in pipe nopipe; do
  f=""; [ $v = nopipe ] && f="--no-pipeline"
  timeout 300 python bench.py --no-cpu-baseline $f > $O/bench_$v.json 2> $O/bench_$v.err
  echo "bench $v rc=$?"; cut -c1-600 $O/bench_$v.json
done
CAGPU_LIB=gym_collision_avoidance_amd/libcagpu_dCSPAD=0.so timeout 300 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_nopipe_cs64.json 2> $O/bench_nopipe_cs64.err
echo "bench nopipe cs64 rc=$?"; cut -c1-400 $O/bench_nopipe_cs64.json
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_driver.json 2> $O/bench_driver.err
cut -c1-400 $O/bench_driver.json
