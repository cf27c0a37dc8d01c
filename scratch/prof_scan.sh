#!/bin/bash
# PMC pass on the config-5 workload (scan_kernel + 50-agent step kernel): separate counter passes, no traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --workload crowd50_laser --steps 20 --warmup 5 --min-warm-seconds 0 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan_stats -- $B > $O/scan_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/scan_sq -- $B > $O/scan_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/scan_sq2 -- $B > $O/scan_sq2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/scan_fetch -- $B > $O/scan_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/scan_write -- $B > $O/scan_write.log 2>&1
find $O -name '*agent_info.csv' -delete
cd $R; python profiles/summarize.py gpurun_out/scan_stats gpurun_out/scan_sq gpurun_out/scan_sq2 gpurun_out/scan_fetch gpurun_out/scan_write | grep -v "at::native\|rocclr" 
