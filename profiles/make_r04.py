#!/usr/bin/env python3
"""profiles/make_r04.py -- copies what scratch/measure_r04.sh left under gpurun_out/r04 into profiles/r04_* (the bench lines,
the per-config table, the HBM traffic and VALU records bench.py quotes, the GA3C rows-vs-time record, the rocprofv3
summary) and prints the counter means the hand-written part of profiles/r04_rocprof_summary.md quotes.  Run in the build
container after the gpurun call."""
import csv
import glob
import json
import os
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "r04")
DST = os.path.join(REPO, "profiles")


def pmc(d, kernel):
    """{counter: mean per step dispatch} -- the first dispatch (cagpu_plan's set-up launch of 1000 waves) left out"""
    acc = {}
    for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in acc.items()}, {k: len(v) - 1 for k, v in acc.items()}


def line(name):
    txt = [l for l in open(os.path.join(SRC, name + ".json")).read().splitlines() if l.startswith("{")]
    return json.loads(txt[-1])


names = ["bench_driver", "bench_n1", "bench_n1_nopipe", "bench_rollout", "bench_2ranks_one_gpu", "cfg2_1024x10", "cfg3_ga3c20",
         "cfg4_32768x10_one_gpu", "cfg5_crowd50"]
lines = {}
for n in names:
    lines[n] = line(n)
    json.dump(lines[n], open(os.path.join(DST, "r04_" + n + ".json"), "w"))
shutil.copy(os.path.join(SRC, "ga3c_rows.json"), os.path.join(DST, "r04_ga3c_rows.json"))

K = "ca_pipe_kernel<10, 4, false>"
fetch, nf = pmc("prof_fetch", K)
write, _ = pmc("prof_write", K)
traffic = {"envs": 4096, "agents": 10, "kernel": "pipe::" + K,
           "fetch_kb_per_launch": round(fetch["FETCH_SIZE"], 1), "write_kb_per_launch": round(write["WRITE_SIZE"], 1),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), gpurun_out/r04 via scratch/measure_r04.sh; "
                     "%d step dispatches (the one cagpu_plan dispatch of the set-up removed from the mean); KB = 1024 B" % nf["FETCH_SIZE"]}
json.dump(traffic, open(os.path.join(DST, "r04_traffic.json"), "w"), indent=1)
sq, nsq = pmc("prof_sq", K)
sq2, _ = pmc("prof_sq2", K)
valu = {"envs": 4096, "agents": 10, "kernel": "pipe::" + K, "valu_insts_per_launch": round(sq["SQ_INSTS_VALU"]),
        "salu_insts_per_launch": round(sq["SQ_INSTS_SALU"]), "lds_insts_per_launch": round(sq["SQ_INSTS_LDS"]),
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the device's SIMDs: / (256 CUs x 4 SIMDs) x 4 = cycles per SIMD
        "valu_busy_cycles_per_simd": sq2["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0,
        "wave_cycles": round(sq["SQ_WAVE_CYCLES"]), "wait_any_cycles": round(sq["SQ_WAIT_ANY"]),
        "source": "rocprofv3 --pmc SQ_INSTS_VALU ... / SQ_ACTIVE_INST_VALU ... (two passes of the same command), gpurun_out/r04 "
                  "via scratch/measure_r04.sh; %d step dispatches" % nsq["SQ_INSTS_VALU"]}
json.dump(valu, open(os.path.join(DST, "r04_valu.json"), "w"), indent=1)

labels = [("bench_n1", "metric: 4096 x 10 RVO, one launch per step"),
          ("bench_driver", "metric, the driver's command (--steps 20 --warmup 5): median of the repeated 20-step blocks"),
          ("bench_rollout", "metric workload, cagpu_rollout (2000 steps per launch)"),
          ("bench_2ranks_one_gpu", "`python bench.py --gpus 2` with no launcher: two self-started ranks sharing ONE GPU over gloo (the N > 1 code path; not a scaling number)"),
          ("cfg2_1024x10", "configs[1]: 1024 envs x 10 agents RVO"),
          ("cfg3_ga3c20", "configs[2]: 4096 x 20 GA3C-CADRL"),
          ("cfg4_32768x10_one_gpu", "configs[3]-shaped on ONE GPU: 32768 x 10 RVO"),
          ("cfg5_crowd50", "configs[4]: 4096 x 50 RVO + map + LaserScanSensor"),
          ("bench_n1_nopipe", "metric without CaState.next_action (the unpipelined round-2 kernel)")]
out = []
for n, lab in labels:
    d = lines[n]
    row = {"config": lab}
    for k in ("value", "unit", "n_gpus", "ranks_seen", "ms_per_step", "event_ms_per_step", "steps", "timed_blocks", "roofline"):
        if k in d:
            row[k] = d[k]
    row["workload"] = d["config"]["workload"]
    for k in ("env_api", "rollout", "cpu_baseline"):
        if k in d:
            row[k] = d[k]
    out.append(row)
json.dump(out, open(os.path.join(DST, "r04_configs.json"), "w"), indent=1)
# (the raw tables of profiles/summarize.py: appended by hand behind the written part of profiles/r04_rocprof_summary.md)
shutil.copy(os.path.join(SRC, "summary.md"), os.path.join(DST, "r04_rocprof_tables.md"))

print("traffic KB", traffic["fetch_kb_per_launch"], traffic["write_kb_per_launch"],
      "MB total %.2f" % ((traffic["fetch_kb_per_launch"] + traffic["write_kb_per_launch"]) * 1024 / 1e6))
for d, k in (("prof_sq", K), ("prof_sq2", K), ("prof_sq_rollout", "ca_pipe_kernel<10, 4, true>")):
    m, n = pmc(d, k)
    print(d, {a: round(b) for a, b in m.items()}, "n", set(n.values()))
for n in names:
    d = lines[n]
    print("%-24s value %.4g ms/step %.5f events %.5f frac %.4f" % (n, d["value"], d["ms_per_step"], d.get("event_ms_per_step", 0), d.get("roofline", {}).get("frac", 0)))
for f in glob.glob(os.path.join(SRC, "prof_*", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r["Name"] for s in ("ca_pipe", "ca_kernel", "ga3c", "scan_kernel", "compact")):
            print(os.path.basename(os.path.dirname(os.path.dirname(f))), r["Name"][:60], r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
