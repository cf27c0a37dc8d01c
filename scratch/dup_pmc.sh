#!/bin/bash
# in-situ cost of the pipelined kernel's phases: instruction counters of the launch with one phase run twice (build variants
# dPIPE_DUP=<bit>,fast) against the plain fast build, all in one call
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dup
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename "$lib" .so)
  P="python $R/bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 500 --min-warm-seconds 0"
  CAGPU_LIB=$R/$lib timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$O/$n" -- $P > "$O/$n.log" 2>&1
done
find $O -name '*agent_info.csv' -delete
cd $R
python - <<'PY'
import csv, glob, os, collections
base = None
for d in sorted(glob.glob("gpurun_out/dup/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ca_pipe_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the steady-state launches: drop the first 100 dispatches
    m = {k: sum(v[100:]) / max(1, len(v[100:])) for k, v in acc.items()}
    name = os.path.basename(d.rstrip("/"))
    print("%-40s VALU %9.0f SALU %9.0f LDS %8.0f wavecycles %10.0f (n=%d)" % (name, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_WAVE_CYCLES", 0), len(acc.get("SQ_INSTS_VALU", []))))
PY
du -sh $O
