#!/bin/bash
# PMC passes over ga3c_kernel (scratch/ga3c_loop.py: 150 steps of the config-3 workload, then 40 back-to-back launches at the
# steady state's ~30 k live rows; the means below run over all 190 dispatches)
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/ga3c_pmc${1:+_$1}   # optional tag: one output directory per build (CAGPU_LIB selects the library)
export PMC_OUT=$O
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="env WARM=150 N=40 python $R/scratch/ga3c_loop.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $P > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/p2 -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_WAIT_ANY --output-format csv -d $O/p3 -- $P > $O/p3.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = os.environ["PMC_OUT"]
for d in ("p1", "p2", "p3"):
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ga3c_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-34s n=%d mean %.4g" % (k, len(v), sum(v) / len(v)))
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
