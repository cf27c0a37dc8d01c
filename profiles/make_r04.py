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


# ---- profiles/r04_rocprof_summary.md: the written head (numbers from the files above) + summarize.py's tables + the
# counter passes over ga3c_kernel
import re


def kstats(d, needle):
    for f in glob.glob(os.path.join(SRC, d, "**", "*kernel_stats.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if needle in r["Name"]]
        rows.sort(key=lambda r: -int(r["Calls"]))
        if rows:
            r = rows[0]
            return {"calls": int(r["Calls"]), "avg": float(r["AverageNs"]) / 1e3, "min": float(r["MinNs"]) / 1e3, "max": float(r["MaxNs"]) / 1e3}
    return {"calls": 0, "avg": float("nan"), "min": float("nan"), "max": float("nan")}


def sp(n):
    return ("%d" % n if n < 10000 else "{:,}".format(n).replace(",", " "))


b, dr, ro = lines["bench_n1"], lines["bench_driver"], lines["bench_rollout"]
c2, c3, c4, c5, r2 = lines["cfg2_1024x10"], lines["cfg3_ga3c20"], lines["cfg4_32768x10_one_gpu"], lines["cfg5_crowd50"], lines["bench_2ranks_one_gpu"]
tb, td = b["timed_blocks"], dr["timed_blocks"]
kp, kg, kc, ks20 = kstats("prof_stats", "ca_pipe_kernel<10, 4, false>"), kstats("prof_ga3c", "ga3c_kernel"), kstats("prof_ga3c", "compact_kernel"), kstats("prof_ga3c", "ca_kernel<256, false, 20")
kscan, ks50 = kstats("prof_crowd", "scan_kernel"), kstats("prof_crowd", "ca_kernel<512")
passed = re.findall(r"(\d+) passed", open(os.path.join(SRC, "pytest_gpu.log")).read())[-1]
soak = re.findall(r"(\d+) passed", open(os.path.join(SRC, "soak.log")).read())
rows_rec = json.load(open(os.path.join(DST, "r04_ga3c_rows.json")))
by = {r["rows"]: r["us"] for r in rows_rec["by_rows"]}
near = lambda n: by[min(by, key=lambda k: abs(k - n))]
sq_r, _ = pmc("prof_sq_rollout", "ca_pipe_kernel<10, 4, true>")
cb = b["cpu_baseline"]
rp = cb.get("reference_python", {})
head = open(os.path.join(DST, "r04_summary_head.md.in")).read().format(
    passed=passed, soak=" x ".join(["%d" % len(soak), soak[0]]) if soak else "?",
    n1_us=b["event_ms_per_step"] * 1e3, n1_val=b["value"] / 1e9, n1_blocks=tb["blocks"], n1_min=tb["ms_per_step_min"] * 1e3, n1_max=tb["ms_per_step_max"] * 1e3,
    kp_avg=kp["avg"], kp_calls=sp(kp["calls"]), kp_min=kp["min"], kp_max=kp["max"],
    dr_wall=dr["ms_per_step"] * 1e3, dr_ev=dr["event_ms_per_step"] * 1e3, dr_val=dr["value"] / 1e9, dr_dev=td["device_seconds_timed"],
    dr_first=td["ms_per_step_first"] * 1e3, dr_min=td["ms_per_step_min"] * 1e3, dr_max=td["ms_per_step_max"] * 1e3,
    n1_gbs=b["roofline"]["achieved"], n1_frac=b["roofline"]["frac"] * 100, dr_frac=dr["roofline"]["frac"] * 100,
    ro_us=ro["ms_per_step"] * 1e3, ro_val=ro["value"] / 1e9, ro_frac=ro["roofline"]["frac"] * 100,
    fetch=traffic["fetch_kb_per_launch"], write=traffic["write_kb_per_launch"],
    traffic_mb=(traffic["fetch_kb_per_launch"] + traffic["write_kb_per_launch"]) * 1024 / 1e6,
    traffic_x=(traffic["fetch_kb_per_launch"] + traffic["write_kb_per_launch"]) * 1024 / (356.0 * 40960),
    valu_m=valu["valu_insts_per_launch"] / 1e6, salu_m=valu["salu_insts_per_launch"] / 1e6, lds_m=valu["lds_insts_per_launch"] / 1e6,
    valu_qc=sq2["SQ_ACTIVE_INST_VALU"] / 1e6, valu_kc=valu["valu_busy_cycles_per_simd"] / 1e3, valu_us=valu["valu_busy_cycles_per_simd"] / 2400.0,
    valu_busy=b["roofline"]["valu"]["valu_busy_frac"] * 100,
    ro_qc=sq_r["SQ_ACTIVE_INST_VALU"] / 1e9, ro_kc=sq_r["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / 300.0 / 1e3,
    ro_valu_us=sq_r["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / 300.0 / 2400.0,
    ro_busy=sq_r["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / 300.0 / 2400.0 / (ro["ms_per_step"] * 1e3) * 100,
    wave_m=valu["wave_cycles"] / 1e6, wait_m=valu["wait_any_cycles"] / 1e6, wait_pct=100.0 * valu["wait_any_cycles"] / valu["wave_cycles"],
    bank_k=sq2["SQ_LDS_BANK_CONFLICT"] / 1e3,
    env_def=b["env_api"]["default"]["us_per_step"], env_zc=b["env_api"]["zero_copy"]["us_per_step"],
    cpu1=cb["single_core_value"] / 1e6, cpuall=cb["value"] / 1e7, cpu_cores=cb["cores"],
    ref1=rp.get("one_process", float("nan")) / 1e3, refall=rp.get("all_cores", float("nan")) / 1e4, ref_cores=rp.get("cores", 0),
    c2_us=c2["ms_per_step"] * 1e3, c2_val=c2["value"] / 1e8, c2_frac=c2["roofline"]["frac"] * 100,
    kc_avg=kc["avg"], kg_avg=kg["avg"], ks20_avg=ks20["avg"], g26=near(26131), g32=near(31352),
    c3_ms=c3["ms_per_step"], c3_val=c3["value"] / 1e8,
    c4_us=c4["ms_per_step"] * 1e3, c4_val=c4["value"] / 1e9, c4_frac=c4["roofline"]["frac"] * 100,
    c5_step=c5["roofline"]["step_kernel_us"] / 1e3, c5_scan=c5["roofline"]["scan_kernel_us"] / 1e3, c5_ms=c5["ms_per_step"], c5_val=c5["value"] / 1e8,
    c5_gbs=c5["roofline"]["achieved"], c5_frac=c5["roofline"]["frac"] * 100, ks50_ms=ks50["avg"] / 1e3, kscan_ms=kscan["avg"] / 1e3,
    r2a=r2["per_rank_event_ms_per_step"][0] * 1e3, r2b=r2["per_rank_event_ms_per_step"][1] * 1e3, r2_eps=r2["episode_stats"]["episodes"] / 1e5)
tables = open(os.path.join(SRC, "summary.md")).read()
tables = re.sub(r"/tmp/code/[^ )]*?/repo/", "", tables)
tables = tables[tables.index("## rocprofv3 --kernel-trace --stats"):]
pm_txt = open(os.path.join(SRC, "ga3c_pmc.txt")).read().splitlines()
cnt = {l.split()[0]: float(l.split()[-1]) for l in pm_txt if l.startswith("SQ_")}
gk = [l for l in pm_txt if "ga3c_kernel" in l and not l.startswith(("W2", "E2", "I2"))]
g_avg = float(gk[-1].split()[-1]) / 1e3 if gk else float("nan")
n_disp = [l for l in pm_txt if l.startswith("SQ_WAVES")][0].split("n=")[1].split()[0]
ga = ("\n## rocprofv3 --pmc over `ga3c::ga3c_kernel` (`scratch/ga3c_pmc.sh`: 150 steps of config 3, then 40 back-to-back launches; "
      "means over all %s dispatches, average launch %.0f us)\n\nThree separate `--pmc` passes (never combined with a trace).  `SQ_VALU_MFMA_BUSY_CYCLES` counts cycles "
      "summed over the 1024 SIMDs: %.3g / 1024 = %.0f k cycles per SIMD of a %.0f us launch (~%.0f k cycles at 2.4 GHz) = the matrix pipe busy "
      "**%.0f %%**; `SQ_ACTIVE_INST_VALU` counts quad-cycles: %.3g x 4 / 1024 = %.0f k cycles = the VALU issuing **%.0f %%**; `SQ_WAIT_INST_ANY` / "
      "`SQ_WAVE_CYCLES` = **%.0f %%** of the wave time waiting on an instruction; %.3g MFMA instructions per launch (%.1f busy cycles each on "
      "average: 16 for `v_mfma_f32_16x16x32_bf16`, 32 for the f32 ones).\n\n```\n%s\n```\n" % (
          n_disp, g_avg, cnt["SQ_VALU_MFMA_BUSY_CYCLES"], cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024e3, g_avg, g_avg * 2.4,
          100 * cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (g_avg * 2400), cnt["SQ_ACTIVE_INST_VALU"], cnt["SQ_ACTIVE_INST_VALU"] * 4 / 1024e3,
          100 * cnt["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (g_avg * 2400), 100 * cnt["SQ_WAIT_INST_ANY"] / cnt["SQ_WAVE_CYCLES"], cnt["SQ_INSTS_MFMA"],
          cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt["SQ_INSTS_MFMA"], "\n".join(l for l in pm_txt if l.startswith("SQ_"))))
open(os.path.join(DST, "r04_rocprof_summary.md"), "w").write(head + tables + ga)
os.remove(os.path.join(DST, "r04_rocprof_tables.md"))
print("wrote profiles/r04_rocprof_summary.md")
