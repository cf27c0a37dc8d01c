#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/scanpmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --workload crowd50_laser --steps 10 --warmup 3 --no-cpu-baseline --min-warm-seconds 0"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- $P > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -- $P > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_BRANCH --output-format csv -d $O/sq3 -- $P > $O/sq3.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R
python profiles/summarize.py $O/sq $O/sq2 $O/sq3 | grep "scan_kernel"
tail -3 $O/sq3.log
