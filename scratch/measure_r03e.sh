#!/bin/bash
# PMC passes of the pipelined step kernel (and the unpipelined one, same call): counters in their own --pmc passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in pipe nopipe; do
  f=""; [ $v = nopipe ] && f="--no-pipeline"
  P="python $R/bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 500 --min-warm-seconds 0 $f"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq_$v -- $P > $O/sq_$v.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/sq2_$v -- $P > $O/sq2_$v.log 2>&1
done
P="python $R/bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 500 --min-warm-seconds 0 --mode rollout"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq_rollout -- $P > $O/sq_rollout.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/sq2_rollout -- $P > $O/sq2_rollout.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R
python profiles/summarize.py $O/sq_pipe $O/sq2_pipe $O/sq_nopipe $O/sq2_nopipe $O/sq_rollout $O/sq2_rollout > $O/summary.md 2>&1
cat $O/summary.md | grep -v "^$" | head -120
du -sh $O
