#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*/) into the small summaries committed under profiles/.

    python profiles/summarize.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write > profiles/rNN_rocprof_summary.md
"""
import collections
import csv
import glob
import os
import sys


def kernel_stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return
    print("## rocprofv3 --kernel-trace --stats (%s)\n" % d)
    print("| kernel | calls | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f[0])):
        name = r["Name"].replace("(anonymous namespace)::", "")[:70]
        print("| `%s` | %s | %.2f | %.2f | %.2f | %s |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                          r["Percentage"]))
    print()


def counters(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if "ca_kernel" in r["Kernel_Name"] or "ca_pipe_kernel" in r["Kernel_Name"] or "ga3c" in r["Kernel_Name"] or "scan_kernel" in r["Kernel_Name"]:
            short = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count",
                                      "Scratch_Size")}
    print("## rocprofv3 --pmc (%s), ca_kernel dispatches only\n" % d)
    print("dispatch geometry:", meta, "\n")
    print("| kernel | counter | dispatches | mean per dispatch | min | max |")
    print("|---|---|---|---|---|---|")
    for (kn, c), v in sorted(agg.items()):
        print("| `%s` | %s | %d | %.1f | %.1f | %.1f |" % (kn, c, len(v), sum(v) / len(v), min(v), max(v)))
    print()


if __name__ == "__main__":
    for d in sys.argv[1:]:
        kernel_stats(d)
        counters(d)
