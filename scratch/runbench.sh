#!/bin/bash
# usage: scratch/runbench.sh  -> tests (-x) + bench step + bench rollout, compact output
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  " | tail -60) > gpurun_out/test_gpu.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/test_gpu.log | tail
for mode in step rollout; do
  (timeout 300 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --mode $mode 2>&1 | tail -1) > gpurun_out/bench_$mode.log
  python -c "
import json;d=json.loads(open('gpurun_out/bench_$mode.log').read());print('$mode', round(d['value']/1e6,1),'M/s',round(d['ms_per_step']*1e3,2),'us/step frac',round(d['roofline']['frac'],4))"
done
