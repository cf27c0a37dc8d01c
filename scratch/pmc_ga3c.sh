#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/pmc_ga3c*
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/pmc_ga3c1 -- python $R/bench.py --workload ga3c20 --steps 20 --warmup 5 > $O/pmc_ga3c1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_ga3c2 -- python $R/bench.py --workload ga3c20 --steps 20 --warmup 5 > $O/pmc_ga3c2.log 2>&1
cd $R
python profiles/summarize.py $O/pmc_ga3c1 $O/pmc_ga3c2 | grep -i "ga3c\|geometry"
tail -3 $O/pmc_ga3c1.log $O/pmc_ga3c2.log | cut -c1-300
find $O -name "*counter_collection.csv" -delete
