#!/usr/bin/env python3
"""profiles/make_r06.py -- files what scratch/measure_r06.sh left under gpurun_out/r06 under profiles/r06_*: the bench lines,
the per-config table, the CALIBRATED HBM-traffic records and the VALU records bench.py quotes (one record per kernel: the
n-step kernel of the look-ahead ring, per step, and the single-step kernel), and profiles/r06_rocprof_summary.md (a generated
head with every number traceable to a csv under gpurun_out/r06 + the raw tables of profiles/summarize.py).  Run in the build
container after the gpurun call."""
import csv
import glob
import json
import os
import re
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "r06")
DST = os.path.join(REPO, "profiles")
E, N, ALG = 4096, 10, 356.0
KM, KS = "ca_pipe_kernel<10, 4, true>", "ca_pipe_kernel<10, 4, false>"   # as cagpu_last_kernel() / the bench lines name them
# ... and as the profiler names the instantiations (round 6: the fourth template parameter is FAIR, the progress-fair priorities)
PROF = {KM: "ca_pipe_kernel<10, 4, true, true>", KS: "ca_pipe_kernel<10, 4, false, false>"}
L_PMC = 20      # steps per launch of the counter passes over the n-step kernel: the length the driver's command times


def _one_run(d, pattern):
    """the csv files of ONE profiler run under gpurun_out/r06/<d>: gpurun merges a call's outputs INTO the local directory, so
    files of an earlier call (another process id in the name) survive unless the directory is cleared first -- averaging them
    in is how a profile goes stale; refuse"""
    files = glob.glob(os.path.join(SRC, d, "**", pattern), recursive=True)
    runs = {os.path.basename(f).split("_")[0] for f in files}
    if len(runs) > 1:
        raise SystemExit("profiles/make_r06.py: %s holds the output of %d profiler runs (%s): rm -rf gpurun_out/r06 before the "
                         "measurement call and run it again" % (d, len(runs), sorted(runs)))
    return files


def pmc(d, kernel, skip=1):
    """{counter: mean per dispatch of `kernel`}, the first `skip` dispatches left out (set-up launches); + their number"""
    acc = {}
    for f in _one_run(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if PROF.get(kernel, kernel) in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v[skip:]) / max(1, len(v) - skip) for k, v in acc.items()}, {k: len(v) - skip for k, v in acc.items()}


def kstats(d, needle):
    for f in _one_run(d, "*kernel_stats.csv"):
        rows = [r for r in csv.DictReader(open(f)) if PROF.get(needle, needle) in r["Name"]]
        rows.sort(key=lambda r: -int(r["Calls"]))
        if rows:
            r = rows[0]
            return {"calls": int(r["Calls"]), "avg": float(r["AverageNs"]) / 1e3, "min": float(r["MinNs"]) / 1e3, "max": float(r["MaxNs"]) / 1e3}
    return {"calls": 0, "avg": float("nan"), "min": float("nan"), "max": float("nan")}


def line(name):
    txt = [l for l in open(os.path.join(SRC, name + ".json")).read().splitlines() if l.startswith("{")]
    return json.loads(txt[-1])


names = ["bench_driver", "bench_n1", "bench_step", "bench_rollout", "bench_lookahead200", "bench_2ranks_one_gpu", "bench_force_dist_rccl",
         "cfg2_1024x10", "cfg2_1024x10_step", "cfg3_ga3c20", "cfg3_ga3c20_fused", "cfg4_32768x10_one_gpu", "cfg4_32768x10_one_gpu_step", "cfg5_crowd50"]
lines = {}
for n in names:
    try:
        lines[n] = line(n)
    except Exception as e:  # noqa: BLE001
        print("missing", n, e)
        continue
    json.dump(lines[n], open(os.path.join(DST, "r06_" + n + ".json"), "w"))

# ---- provenance: every line of this record was measured on ONE library (a profile must not go
# stale unnoticed).  The A/B lines (ab_*) name other libraries on purpose and are filed as a text table with their hashes.
prov = json.load(open(os.path.join(SRC, "provenance.json")))
bad = {n: d.get("provenance", {}).get("lib_sha256") for n, d in lines.items()
       if d.get("provenance", {}).get("lib_sha256") != prov["lib_sha256"] or d.get("provenance", {}).get("host_py_sha256") != prov.get("host_py_sha256")}
if bad:
    raise SystemExit("profiles/make_r06.py: lines measured on another library than %s: %s" % (prov["lib_sha256"], bad))
print("provenance: lib %s  git %s%s  sources %s" % (prov["lib_sha256"][:16], (prov.get("git_sha") or "?")[:12], " (dirty)" if prov.get("git_dirty") else "", (prov.get("source_sha256") or "?")[:16]))
PROV = {"lib_sha256": prov["lib_sha256"], "git_sha": prov.get("git_sha"), "git_dirty": prov.get("git_dirty"), "source_sha256": prov.get("source_sha256"),
        "host_py_sha256": prov.get("host_py_sha256")}
json.dump(PROV, open(os.path.join(DST, "r06_provenance.json"), "w"), indent=1)

# ---- counter calibration: cagpu_debug_copy8 moves exactly 8 n bytes each way with the step kernels' access shape
cal = {}
for d, c in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    for f in _one_run(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "copy8_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                n_el = 5_000_000 if int(r["Grid_Size"]) < 10_000_000 else 50_000_000
                cal.setdefault((c, n_el), []).append(float(r["Counter_Value"]))
factor = {}
for (c, n_el), v in sorted(cal.items()):
    kb = sum(v) / len(v)
    factor[(c, n_el)] = 8.0 * n_el / (kb * 1024.0)
    print("calibration %s n=%d: counter %.1f KB for %d bytes -> factor %.4f" % (c, n_el, kb, 8 * n_el, factor[(c, n_el)]))
f_fetch = factor.get(("FETCH_SIZE", 5_000_000), 2.0)
f_write = factor.get(("WRITE_SIZE", 5_000_000), 1.0)

traffic = []
for kern, dirs, per in ((KM, ("prof_fetch", "prof_write"), L_PMC), (KS, ("prof_fetch_step", "prof_write_step"), 1)):
    fe, nf = pmc(dirs[0], kern)
    wr, _ = pmc(dirs[1], kern)
    if "FETCH_SIZE" not in fe:
        continue
    raw_f, raw_w = fe["FETCH_SIZE"], wr["WRITE_SIZE"]
    tb = (raw_f * f_fetch + raw_w * f_write) * 1024.0 / per
    traffic.append({"envs": E, "agents": N, "kernel": kern, "steps_per_launch": per, "provenance": PROV,
                    "traffic_bytes_per_launch": round(tb * per),
                    "fetch_kb_per_launch_raw": round(raw_f, 1), "write_kb_per_launch_raw": round(raw_w, 1),
                    "fetch_factor": round(f_fetch, 4), "write_factor": round(f_write, 4),
                    "traffic_bytes_per_step": round(tb), "traffic_over_algorithmic": round(tb / (ALG * E * N), 3),
                    "calibration": "cagpu_debug_copy8 (8 B per lane and instruction, 40 MB and 400 MB each way) under the same "
                                   "--pmc passes: FETCH_SIZE reports 1 / %.3f of the bytes read, WRITE_SIZE 1 / %.3f of the bytes "
                                   "written (scratch/copy8_calib.py; gpurun_out/r06/calib_*)" % (f_fetch, f_write),
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), scratch/measure_r06.sh; %d dispatches; KB = 1024 B" % nf["FETCH_SIZE"]})
if len(traffic) != 2:
    raise SystemExit("profiles/make_r06.py: no counter rows for %s / %s under gpurun_out/r06 (kernel names changed?)" % (PROF[KM], PROF[KS]))
json.dump(traffic, open(os.path.join(DST, "r06_traffic.json"), "w"), indent=1)

valu = []
sq, nsq = pmc("prof_sq", KM)
sq2, _ = pmc("prof_sq2", KM)
if "SQ_INSTS_VALU" in sq:
    valu.append({"envs": E, "agents": N, "kernel": KM, "steps_per_launch": L_PMC, "provenance": PROV,
                 "valu_insts_per_launch": round(sq["SQ_INSTS_VALU"]), "salu_insts_per_launch": round(sq["SQ_INSTS_SALU"]),
                 "lds_insts_per_launch": round(sq["SQ_INSTS_LDS"]),
                 # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the device's SIMDs: / (256 CUs x 4 SIMDs) x 4 = cycles per SIMD
                 "valu_busy_cycles_per_simd": sq2["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0,
                 "wave_cycles": round(sq["SQ_WAVE_CYCLES"]), "wait_any_cycles": round(sq["SQ_WAIT_ANY"]),
                 "wait_inst_any_cycles": round(sq2.get("SQ_WAIT_INST_ANY", 0)), "lds_bank_conflict_cycles": round(sq2.get("SQ_LDS_BANK_CONFLICT", 0)),
                 "source": "rocprofv3 --pmc (two passes), scratch/measure_r06.sh; %d dispatches of %d steps" % (nsq["SQ_INSTS_VALU"], L_PMC)})
sqs, nss = pmc("prof_sq_step", KS)
if "SQ_INSTS_VALU" in sqs:
    valu.append({"envs": E, "agents": N, "kernel": KS, "steps_per_launch": 1, "provenance": PROV,
                 "valu_insts_per_launch": round(sqs["SQ_INSTS_VALU"]),
                 "valu_busy_cycles_per_simd": sqs["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0,
                 "wave_cycles": round(sqs["SQ_WAVE_CYCLES"]), "wait_any_cycles": round(sqs["SQ_WAIT_ANY"]),
                 "source": "rocprofv3 --pmc (one pass), scratch/measure_r06.sh; %d dispatches" % nss["SQ_INSTS_VALU"]})
json.dump(valu, open(os.path.join(DST, "r06_valu.json"), "w"), indent=1)

labels = [("bench_driver", "metric, the driver's command (--steps 20 --warmup 5): look-ahead ring of 20, median of the repeated 20-step blocks"),
          ("bench_n1", "metric: 4096 x 10 RVO, default (2000 steps: look-ahead ring of 50)"),
          ("bench_lookahead200", "metric workload, look-ahead ring of 200"),
          ("bench_step", "metric workload, one cagpu_step launch per step (round 4's headline path)"),
          ("bench_rollout", "metric workload, cagpu_rollout (2000 steps per launch, only the last step's outputs kept)"),
          ("bench_2ranks_one_gpu", "`python bench.py --gpus 2` with no launcher: two self-started ranks sharing ONE GPU over gloo (the N > 1 code path; not a scaling number)"),
          ("bench_force_dist_rccl", "`bench.py --force-dist`: N = 1 with a one-rank RCCL process group, every collective of the N > 1 path executed"),
          ("cfg2_1024x10", "configs[1]: 1024 envs x 10 agents RVO, look-ahead ring of 64"),
          ("cfg2_1024x10_step", "configs[1], one launch per step"),
          ("cfg3_ga3c20", "configs[2]: 4096 x 20 GA3C-CADRL"),
          ("cfg3_ga3c20_fused", "configs[2] as worded (sensing fused into the network kernel: cagpu_ga3c with obs = NULL)"),
          ("cfg4_32768x10_one_gpu", "configs[3]-shaped on ONE GPU: 32768 x 10 RVO, look-ahead ring of 50"),
          ("cfg4_32768x10_one_gpu_step", "configs[3]-shaped on ONE GPU, one launch per step"),
          ("cfg5_crowd50", "configs[4]: 4096 x 50 RVO + map + LaserScanSensor")]
out = []
for n, lab in labels:
    if n not in lines:
        continue
    d = lines[n]
    row = {"config": lab}
    for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "distributed", "ms_per_step", "event_ms_per_step", "steps", "timed_blocks", "roofline"):
        if k in d:
            row[k] = d[k]
    row["workload"], row["launch_mode"] = d["config"]["workload"], d["config"].get("launch_mode")
    for k in ("single_launch", "env_api", "rollout", "cpu_baseline"):
        if k in d:
            row[k] = d[k]
    out.append(row)
json.dump(out, open(os.path.join(DST, "r06_configs.json"), "w"), indent=1)

# ---- the generated summary
kp_m, kp_drv, kp_s = kstats("prof_stats", KM), kstats("prof_stats_driver", KM), kstats("prof_stats_step", KS)
kg, kc, ks20 = kstats("prof_ga3c", "ga3c_kernel"), kstats("prof_ga3c", "compact_kernel"), kstats("prof_ga3c", "ca_kernel<256, false, 20")
kscan, ks50 = kstats("prof_crowd", "scan_kernel"), kstats("prof_crowd", "ca_kernel<512")
passed = re.findall(r"(\d+) passed", open(os.path.join(SRC, "pytest_gpu.log")).read())
b, dr = lines["bench_n1"], lines["bench_driver"]
md = []
md.append("# Round 6: rocprofv3 summary (generated by profiles/make_r06.py from gpurun_out/r06, scratch/measure_r06.sh)\n")
md.append("Measured on: library sha256 `%s`, built at git `%s`%s, kernel sources `%s` (profiles/r06_provenance.json; every bench line under "
          "profiles/r06_*.json carries the same `provenance` block, this script refuses lines of another library).\n" % (
              PROV["lib_sha256"], (PROV["git_sha"] or "?")[:12], " + uncommitted changes" if PROV["git_dirty"] else "", (PROV["source_sha256"] or "?")[:16]))
md.append("GPU test-suite on the measured tree: **%s passed**.\n" % (passed[-1] if passed else "?"))
md.append("| line | launch mode | agent-steps/s | us / step wall | us / step events | kernel | avg launch us (events) | frac of HBM |")
md.append("|---|---|---|---|---|---|---|---|")
for n, lab in labels:
    if n not in lines:
        continue
    d = lines[n]
    r = d["roofline"]
    md.append("| %s | %s | %.3e | %.3f | %.3f | `%s` | %.1f | %.4f |" % (lab.split(":")[0][:60], d["config"].get("launch_mode"), d["value"], d["ms_per_step"] * 1e3,
                                                                      d.get("event_ms_per_step", 0) * 1e3, r.get("kernel", "")[:44], r.get("avg_launch_us", 0), r.get("frac", 0)))
md.append("")
md.append("Kernel trace (`rocprofv3 --kernel-trace --stats`) against the bench lines' HIP events: `%s` default run (50 steps per launch) "
          "%d dispatches, average %.1f us = **%.3f us per step** (events: %.3f); the driver's command (20 steps per launch) %d dispatches, "
          "average %.1f us = **%.3f us per step** (events: %.3f); `%s` (one launch per step) %d dispatches, average **%.2f us** (events: %.2f).\n" % (
              KM, kp_m["calls"], kp_m["avg"], kp_m["avg"] / 50.0, b["roofline"]["us_per_step"], kp_drv["calls"], kp_drv["avg"], kp_drv["avg"] / 20.0,
              dr["roofline"]["us_per_step"], KS, kp_s["calls"], kp_s["avg"], lines["bench_step"]["roofline"]["us_per_step"] if "bench_step" in lines else float("nan")))
for t in traffic:
    md.append("HBM traffic of `%s` per STEP: FETCH_SIZE %.1f KB x %.3f + WRITE_SIZE %.1f KB x %.3f per launch of %d step(s) = **%.2f MB per step = %.2f x the "
              "algorithmic 14.58 MB** (counters calibrated on `cagpu_debug_copy8`: %s).\n" % (
                  t["kernel"], t["fetch_kb_per_launch_raw"], t["fetch_factor"], t["write_kb_per_launch_raw"], t["write_factor"], t["steps_per_launch"],
                  t["traffic_bytes_per_step"] / 1e6, t["traffic_over_algorithmic"], t["calibration"]))
for v in valu:
    per = v["steps_per_launch"]
    md.append("VALU of `%s` per step: %.2f M instructions, busy %.0f cycles per SIMD (SQ_ACTIVE_INST_VALU x 4 / 1024) = %.2f cycles per wave-instruction; "
              "SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.0f %%.\n" % (v["kernel"], v["valu_insts_per_launch"] / per / 1e6, v["valu_busy_cycles_per_simd"] / per,
                                                             v["valu_busy_cycles_per_simd"] * 1024.0 / v["valu_insts_per_launch"],
                                                             100.0 * v["wait_any_cycles"] / max(1, v["wave_cycles"])))
md.append("Config 3 kernels (profiler averages over the bench run): compaction %.1f + network %.1f + step kernel %.1f us; config 5: step kernel %.3f + scan kernel %.3f ms.\n" % (
    kc["avg"], kg["avg"], ks20["avg"], ks50["avg"] / 1e3, kscan["avg"] / 1e3))
ab = []
for v in ("product", "dPIPE_YIELD_T=0", "r06end_fast"):
    row = []
    for m in ("l20", "l50", "ro"):
        f = os.path.join(SRC, "ab_%s_%s.json" % (m, v.replace("r06end", "r05end")))
        try:
            d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
            row.append((d["event_ms_per_step"] * 1e3, d["ms_per_step"] * 1e3, d["provenance"]["lib_sha256"][:12]))
        except Exception:  # noqa: BLE001
            row.append(None)
    if any(row):
        ab.append((v.replace("r06end", "r05end"), row))
if ab:
    md.append("Same-box A/B of the n-step kernel's progress-fair priorities (us per step by HIP events / wall; ring of 20 = the driver's command, ring of 50, 2000-step rollout):\n")
    md.append("| library | sha256 | ring of 20 | ring of 50 | rollout |")
    md.append("|---|---|---|---|---|")
    for v, row in ab:
        md.append("| %s | %s | %s |" % ({"product": "this round's product (yield_t = 2, level 1)", "dPIPE_YIELD_T=0": "the same source, -DCAGPU_PIPE_YIELD_T=0",
                                        "r05end_fast": "round 5's end state (commit 4df360a, N = 10 instantiations)"}[v],
                                       next((r[2] for r in row if r), "?"), " | ".join("%.3f / %.3f" % (r[0], r[1]) if r else "-" for r in row)))
    md.append("")
tables = open(os.path.join(SRC, "summary.md")).read()
tables = re.sub(r"/tmp/code/[^ )]*?/repo/", "", tables)
if "## rocprofv3 --kernel-trace --stats" in tables:
    tables = tables[tables.index("## rocprofv3 --kernel-trace --stats"):]
open(os.path.join(DST, "r06_rocprof_summary.md"), "w").write("\n".join(md) + "\n" + tables)
for n in names:
    if n in lines:
        d = lines[n]
        print("%-28s %-13s value %.4g us/step %.3f events %.3f frac %.4f" % (n, d["config"].get("launch_mode"), d["value"], d["ms_per_step"] * 1e3,
                                                                            d.get("event_ms_per_step", 0) * 1e3, d.get("roofline", {}).get("frac", 0)))
print("wrote profiles/r06_*")
