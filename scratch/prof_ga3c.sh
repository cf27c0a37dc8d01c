#!/bin/bash
# HBM traffic of the config-3 kernels (does the observation tensor the step kernel wrote reach ga3c_kernel from HBM or from the Infinity Cache?)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --workload ga3c20 --steps 30 --warmup 5 --min-warm-seconds 0 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/g_fetch -- $B > $O/g_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/g_write -- $B > $O/g_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/g_sq -- $B > $O/g_sq.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R; python profiles/summarize.py gpurun_out/g_fetch gpurun_out/g_write gpurun_out/g_sq | grep -v "at::native\|rocclr"
