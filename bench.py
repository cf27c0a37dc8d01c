#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the batched collision-avoidance hot path on MI355X.

Metric (BASELINE.json): agent-steps/sec at 4096 envs x 10 agents, RVOPolicy (ORCA) + UnicycleDynamics +
OtherAgentsStatesSensor (K=9, closest_first), EvaluateConfig constants (DT=0.1, MAX_TIME_RATIO=8), fixture
cases 10_agents_500_cases with deterministic auto-reset.  One "step" = one `env.step(None)` of every env of the
shard, every step's observations / rewards / done flags handed to the caller.  Default launch mode "lookahead": the
product path of `CollisionAvoidanceEnv.step(None)` -- the next L steps computed in ONE launch of the fused n-step kernel
(cagpu_rollout_ring through the C ABI) and served slot by slot (core.BatchedSim.step_lookahead; bit-identical to one
launch per step, tests/test_gpu_ring.py); L = the largest divisor of --steps up to 64, so a timed block of K steps is
exactly K / L launches and nothing is computed that is not counted.  `--mode step` = one cagpu_step launch per step
(what a caller with external actions gets), reported beside the headline as `single_launch`.  Weak scaling: every GPU
owns 4096 envs (config 4 = 8 x 4096); the only collective is one RCCL all-reduce of the 8 episode counters.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--mode lookahead|step|graph|rollout] [--workload rvo10|ga3c20|crowd50_laser]
The default workload is the metric's; `ga3c20` (BASELINE config 3: 4096 x 20 GA3C-CADRL agents, network on the fp32
matrix cores) and `crowd50_laser` (config 5: 4096 x 50 RVO agents + static map + LaserScanSensor) are the "next" rows,
measured with the same harness and reported with their own roofline (profiles/).
N>1: either the driver's launcher form  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
or plain  python bench.py --gpus N  (no WORLD_SIZE in the environment): bench.py then starts the N ranks itself with that
same launcher.  Either way one rank per GPU; the line is refused unless exactly N ranks took part.  Rank 0 prints ONE
JSON line.  The timed block of EXACTLY K steps is repeated until >= 0.5 s of device time has been measured and the
MEDIAN block is reported (`timed_blocks` carries first / min / max).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32, dense (MI355X_MICROARCH.md)
# GA3C-CADRL network, multiply-adds per query with all 19 LSTM steps live
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_16x16x32_f16 / _bf16 (and 32x32x16), dense (MI355X_MICROARCH.md)
GA3C_MACS = 19 * 71 * 256 + 68 * 256 + 2 * 256 * 256 + 256 * 11
# of which on the f16 matrix cores as THREE plane products each (float32 operands as two fp16 planes, hi = fp16(x), lo =
# fp16(x - hi): 22 of the 24 significant bits; rounds 2 - 4: three bf16 planes, six products): the whole LSTM (x_t and the bias
# too -- K = 32 as four groups of 8, one MFMA per gate), the h part of layer1, layer2, fullyconnected1; the rest (layer1's host
# inputs, logits) on the exact f32 MFMA
GA3C_MACS_BF16 = 19 * 71 * 256 + 64 * 256 + 2 * 256 * 256
GA3C_PLANE_PRODUCTS = 3


def provenance():
    """what this line was measured on: the library file's sha256 (whatever CAGPU_LIB points to), the build record written
    beside the product library (git commit + dirty flag + digest of the kernel sources: gym_collision_avoidance_amd/
    build_native.write_build_info -- there is no .git on the GPU box), this script's own hash"""
    import hashlib
    from gym_collision_avoidance_amd import _native as nat
    from gym_collision_avoidance_amd import build_native as bn
    host = hashlib.sha256()   # the host side of the path: the binding and the batched-state layer (ring bookkeeping, probes)
    for f in ("_native.py", "core.py", "sharding.py"):
        host.update(f.encode() + b"\0" + open(os.path.join(REPO, "gym_collision_avoidance_amd", f), "rb").read())
    out = {"lib": os.path.relpath(nat.LIB_PATH, REPO), "lib_sha256": bn.file_sha256(nat.LIB_PATH) if os.path.exists(nat.LIB_PATH) else None,
           "bench_py_sha256": hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest(), "host_py_sha256": host.hexdigest()}
    info = bn.build_info()
    if os.path.abspath(nat.LIB_PATH) == os.path.abspath(bn.OUT) and not info.get("stale"):
        out.update({k: info.get(k) for k in ("git_sha", "git_dirty", "source_sha256")})
    else:
        out["note"] = "not the product library of the build record (an experiment build through CAGPU_LIB, or a stale record)"
    return out


def _profile_records(stem):
    """the records of profiles/r*_<stem>.json, newest round first (a file holds one record or a list of them)"""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s.json" % stem)), reverse=True):
        d = json.load(open(f))
        for r in (d if isinstance(d, list) else [d]):
            out.append((r, "profiles/" + os.path.basename(f)))
    return out


def measured_traffic_bytes(envs, agents, kernel, steps_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate passes, each
    corrected by the factor the same pass measured on a streaming copy of known size with this kernel's 8-byte-per-lane
    access shape -- cagpu_debug_copy8; profiles/), if they were taken at this geometry and for this kernel; bench.py itself
    does not run the profiler.  -> (bytes per launch or None, source or None)"""
    recs = [(d, src) for d, src in _profile_records("traffic")
            if d.get("envs") == envs and d.get("agents") == agents and d.get("kernel", "").split("(")[0] in kernel]
    for d, src in recs:   # (newest round first) a pass over launches of exactly this length
        if "traffic_bytes_per_launch" in d and d.get("steps_per_launch") == steps_per_launch:
            return d["traffic_bytes_per_launch"], src + " (counter passes over launches of %d steps)" % steps_per_launch
    for d, src in recs:   # otherwise: the per-step figure of a pass at another launch length, scaled -- and said so
        if "traffic_bytes_per_step" in d:
            return d["traffic_bytes_per_step"] * steps_per_launch, src + " (measured at %d steps per launch, scaled to %d)" % (
                d.get("steps_per_launch", 1), steps_per_launch)
        if steps_per_launch == 1 and "fetch_kb_per_launch" in d:   # (records of rounds 2 - 4: raw counters of single-step launches)
            return (d["fetch_kb_per_launch"] + d["write_kb_per_launch"]) * 1024.0, src + " (uncalibrated counters)"
    return None, None


def algorithmic_bytes_per_agent_step(K):
    """read 44 B + write state 28 B + outputs (32 + 28 K) B = 104 + 28 K (356 B at K=9)."""
    return 104 + 28 * K


def _port_leg(args):
    """one process of the CPU port: E envs x n_agents stepped for ~budget_s seconds (module level: picklable)"""
    n_agents, K, budget_s, rank = args
    from oracle import ca_oracle as orc
    table = np.load(os.path.join(REPO, "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n%d" % n_agents]
    E = 64
    o = orc.Oracle(orc.default_params(E, n_agents, max_obs=K))
    o.s["policy"][:] = orc.POL_RVO
    o.reset(table[(np.arange(E) + 64 * rank) % table.shape[0]])
    t0 = time.perf_counter()
    o.rollout(table, 50)
    dt = time.perf_counter() - t0
    steps = min(max(50, int(budget_s / max(dt / 50, 1e-9))), 200000)
    t0 = time.perf_counter()
    o.rollout(table, steps)
    dt = time.perf_counter() - t0
    return E * n_agents * steps, dt


def cpu_baseline(n_agents, K, budget_s=8.0):
    """CPU baseline beside the GPU number, on THIS host (the GPU box), bounded to ~20 s:
      * kind "port": oracle/ca_oracle.cpp (C++ restatement of the reference step) on 1 core and on all cores
        (independent processes, rates summed -- envs never interact);
      * the reference's OWN Python env.step cannot run here: a Python reference may not travel to the GPU box in any form
        (source, bytecode or otherwise -- the task's rule for Python references; only C / C++ references may be built into
        oracle/_ref), so its rate, measured in the build container by oracle/time_reference.py on the unmodified
        /root/reference (1 process and nproc processes, host stated), is attached from the newest
        profiles/r*_reference_cpu.json and re-timed live wherever the reference is present."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    s1, d1 = _port_leg((n_agents, K, budget_s, 0))
    allv, procs = None, cores          # one process per LOGICAL core of this host
    try:
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(_port_leg, [(n_agents, K, budget_s, r) for r in range(procs)])
        allv = sum(s / d for s, d in res)
    except Exception as e:  # noqa: BLE001 -- the baseline must never take the bench line down
        allv, procs = None, 0
        sys.stderr.write("cpu_baseline all-cores leg failed: %r\n" % (e,))
    out = {"value": allv if allv else s1 / d1, "unit": "agent-steps/s", "cores": procs if allv else 1, "kind": "port",
           "host_logical_cores": cores, "single_core_value": s1 / d1,
           "sample": "C++ oracle (oracle/ca_oracle.cpp), 64 envs x %d agents per process, fixture cases with auto-reset: "
                     "1 process for %.1f s (%d agent-steps), then %d processes x ~%.0f s" % (n_agents, d1, s1, procs, budget_s)}
    out["reference_python"] = reference_python_rate()
    return out


def reference_python_rate():
    """The reference's own Python env.step.  Where /root/reference exists (the build container) it is RE-TIMED now by
    oracle/time_reference.py (bounded: ~5 s on one core + ~5 s on all cores); on the GPU box, where the reference cannot
    travel, the newest committed record profiles/r*_reference_cpu.json (written by the same script) is attached."""
    import glob
    import subprocess
    live = None
    if os.path.isdir(os.environ.get("CA_REFERENCE_ROOT", "/root/reference")):
        try:
            tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cagpu_reference_cpu_%d.json" % os.getpid())
            subprocess.run([sys.executable, os.path.join(REPO, "oracle", "time_reference.py"), "--seconds", "5", "--out", tmp],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            live = json.load(open(tmp))
            os.remove(tmp)
        except Exception as e:  # noqa: BLE001 -- the baseline must never take the bench line down
            sys.stderr.write("reference re-timing failed: %r\n" % (e,))
    recs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_reference_cpu.json")))
    if live is None and not recs:
        return None
    r = live if live is not None else json.load(open(recs[-1]))
    return {"kind": "reference",
            "measured_where": "this host, in this run" if live is not None else
                              "build container (the reference does not travel to the GPU box); record profiles/%s" % os.path.basename(recs[-1]),
            "host": r["host"], "one_process": r["one_process"]["agent_steps_per_s"],
            "all_cores": r["all_cores"]["agent_steps_per_s"], "cores": r["all_cores"]["cores"],
            "script": "oracle/time_reference.py"}


def env_api_rates(E, N, steps, torch, dev):
    """What a drop-in user calls: CollisionAvoidanceEnv(num_envs=E).step(None) through the gym-level API of
    gym_collision_avoidance_amd.envs.  `lookahead` = the default env (step(None) served from the look-ahead ring, every
    refill a fresh ring: what step() returned stays the caller's); `single_launch` = lookahead=0 (one launch per step into
    fresh output tensors); `zero_copy` = lookahead=0 with the persistent device buffers.  Host wall clock per step, device
    idle at start and end."""
    os.environ.setdefault("GYM_CONFIG_CLASS", "EvaluateConfig")
    out = {}
    try:
        from gym_collision_avoidance_amd.envs import Config
        from gym_collision_avoidance_amd.envs.collision_avoidance_env import CollisionAvoidanceEnv
        Config.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
        Config.MAX_NUM_OTHER_AGENTS_OBSERVED = N - 1
        for name, kw in (("lookahead", {}), ("single_launch", {"lookahead": 0}), ("zero_copy", {"zero_copy": True})):
            env = CollisionAvoidanceEnv(num_envs=E, device=str(dev), **kw)
            env.set_fixture_suite(N)
            env.reset()
            la = env._sim._la
            ring = la["n"] if la is not None else 0
            # (the ring variant: the adaptive ring starts at 8 steps and doubles -- warm up until it has reached its full length,
            # two more WHOLE rings -- the allocator then holds the blocks the fresh rings alternate between --, stand at a ring
            # boundary, and time whole rings, so that exactly the steps that are served are computed inside the clock)
            n_t = max(2, steps // ring) * ring if ring else steps
            for _ in range(128):
                env.step(None)
            if ring:
                guard = 0
                while (la["len"] != ring or la["t"] < la["len"]) and guard < 16 * ring:
                    env.step(None)
                    guard += 1
                for _ in range(2 * ring):
                    env.step(None)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n_t):
                env.step(None)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            out[name] = {"value": E * N * n_t / dt, "unit": "agent-steps/s", "us_per_step": dt / n_t * 1e6, "steps": n_t}
            if name == "lookahead":
                out[name]["ring"] = ring
            del env
        out["note"] = ("CollisionAvoidanceEnv(num_envs=%d).step(None), ~%d steps (whole rings), host wall clock; lookahead = the default (the longest ring "
                       "CollisionAvoidanceEnv.LOOKAHEAD_BYTES of outputs allow, at most LOOKAHEAD_MAX steps per launch), single_launch = lookahead=0, fresh obs / reward / "
                       "game_over tensors per step, zero_copy = the persistent device buffers (one launch per step)" % (E, steps))
    except Exception as e:  # noqa: BLE001 -- an extra must never take the bench line down
        out["error"] = repr(e)
    return out


def _time_launches(fn, n, torch, dev):
    """average duration of n back-to-back launches of fn on torch's current stream (HIP events)"""
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e-3 / n


def extra_workload(out, a, sim, core, E, N, K, dev, torch):
    """metric / config / roofline of the 'next'-row workloads (not the driver's default line)"""
    n = max(20, min(a.steps, 200))
    if a.workload == "ga3c20":
        infer_s = _time_launches(lambda: sim.ga3c(), n, torch, dev)
        rows = sim.ga3c_rows()   # agents evaluated by the timed launches (live GA3C-CADRL agents, packed by cagpu_ga3c)
        flops = 2.0 * GA3C_MACS * rows
        out["metric"] = "agent-steps/sec at 4096 envs x 20 agents (GA3C-CADRL)"
        out["dtype"] = ("f32 network: float32 operands as two fp16 planes (hi + lo = 22 significant bits, fp16 denormals kept), 3 of the 4 "
                        "plane products on the f16 matrix cores (v_mfma_f32_16x16x32_f16, f32 accumulate; a product within 2^-21), layer1's host inputs "
                        "and the logits on the exact f32 MFMA; f64 simulator state")
        out["config"]["workload"] = ("configs[2]: %d envs/GPU x %d agents, GA3CCADRLPolicy (IROS18 checkpoint, LSTM-64 + "
                                     "3 x FC-256, argmax of 11 actions) + UnicycleDynamics + OtherAgentsStatesSensor K=19 "
                                     "closest_last, fixture n20 (reference generator, seed 0), auto-reset; one cagpu_ga3c "
                                     "+ one cagpu_step launch per step%s" % (E, N, "; FUSED sensing: the network kernel computes its "
                                     "observation rows from the state (obs = NULL)" if a.ga3c_fused else ""))
        # the roofline of a float32 contraction done this way: every product costs three f16 plane products
        peak = BF16_MFMA_PEAK_TFLOPS / GA3C_PLANE_PRODUCTS
        issued = 2.0 * rows * (GA3C_MACS_BF16 * GA3C_PLANE_PRODUCTS) / infer_s / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": flops / infer_s / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": flops / infer_s / 1e12 / BF16_MFMA_PEAK_TFLOPS, "traffic": None,
                           "frac_of_f16_peak": {"algorithmic": flops / infer_s / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                                "issued": issued / BF16_MFMA_PEAK_TFLOPS},
                           "peak_note": "`frac` = algorithmic float32-class flops / the f16 dense MFMA peak (2500 TFLOP/s, MI355X_MICROARCH.md); "
                                        "`issued` counts the three f16 plane products every float32 product costs on this path "
                                        "(rounds 2 - 4: six bf16 ones); against peak / 3 -- the roofline of a float32 contraction done "
                                        "this way -- the algorithmic figure is `frac_of_peak_over_3`",
                           "frac_of_peak_over_3": flops / infer_s / 1e12 / peak,
                           "issued_f16_tflops": issued, "issued_frac_of_f16_peak": issued / BF16_MFMA_PEAK_TFLOPS,
                           "frac_of_f32_mfma_peak": flops / infer_s / 1e12 / F32_MFMA_PEAK_TFLOPS,
                           "kernel": "ga3c::ga3c_kernel", "avg_launch_us": infer_s * 1e6,
                           "algorithmic_flops_per_launch": flops, "macs_per_agent_query": GA3C_MACS,
                           "rows_evaluated": rows, "rows_total": E * N,
                           "note": "flops counted for the %d agents that need an action (not done; the reference queries no "
                                   "others), 19 live LSTM steps each (every agent of a 20-agent env observes 19 others); "
                                   "cagpu_ga3c packs those rows first (CaNet.rows_scratch)" % rows}
    else:
        step_s = _time_launches(lambda: sim.step(), n, torch, dev)
        scan_s = _time_launches(lambda: sim.laserscan(), n, torch, dev)
        b = (104 + 28 * K + 3 * 512 * 4) * E * N
        out["metric"] = "agent-steps/sec at 4096 envs x 50 agents (RVO) + LaserScanSensor"
        out["config"]["workload"] = ("configs[4]: %d envs/GPU x %d agents dense crowd, RVOPolicy + static map (160 x 160 "
                                     "cells, wall collisions) + LaserScanSensor 512 beams x 60 samples x 3 scans + "
                                     "OtherAgentsStatesSensor K=%d, fixture n50 (make_testcase_huge, seed 0), auto-reset; "
                                     "one cagpu_step_map + one cagpu_laserscan launch per step" % (E, N, K))
        out["roofline"] = {"bound": "hbm", "achieved": b / (step_s + scan_s) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": b / (step_s + scan_s) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                           "kernel": "ca_kernel + scan_kernel", "avg_launch_us": (step_s + scan_s) * 1e6,
                           "step_kernel_us": step_s * 1e6, "scan_kernel_us": scan_s * 1e6,
                           "algorithmic_bytes_per_launch": b,
                           "algorithmic_bytes_per_agent_step": 104 + 28 * K + 3 * 512 * 4}


def build_workload(workload, E, dev, rank=0, world=1, pipeline=True, agents=None):
    """The simulator of a bench workload, exactly as the timed run uses it (tests/test_gpu_bench_geometry.py builds its
    full-size parity cases through this function, so the geometry the oracle checks IS the one the bench line reports).
    -> (sim, fixture table, N, K)"""
    from gym_collision_avoidance_amd import _native as nat
    from gym_collision_avoidance_amd import core
    from gym_collision_avoidance_amd.sharding import shard_env_ids
    N = agents if agents else {"rvo10": 10, "ga3c20": 20, "crowd50_laser": 50}[workload]
    K = 19 if workload == "ga3c20" else N - 1
    table = np.load(os.path.join(REPO, "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n%d" % N]
    sort = nat.SORT_CLOSEST_LAST if workload == "ga3c20" else nat.SORT_CLOSEST_FIRST
    sim = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=sort), device=dev, pipeline=pipeline)
    sim.set_plugins(nat.POL_GA3C_CADRL if workload == "ga3c20" else nat.POL_RVO, nat.DYN_UNICYCLE)
    if workload == "ga3c20":
        sim.load_ga3c()
    if workload == "crowd50_laser":  # Map(16 m, 16 m, 0.1 m) with a few wall segments + the 512-beam scan
        sim.set_map(crowd_map())
    off, stride = shard_env_ids(rank, world, E)
    sim.set_fixture_table(table, env_id_offset=off, case_stride=stride)
    sim.reset_from_table()
    return sim, table, N, K


def crowd_map():
    grid = np.zeros((160, 160), dtype=bool)
    grid[40:44, 30:130] = True
    grid[116:120, 30:130] = True
    grid[60:100, 78:82] = True
    return grid


def valu_block(E, N, kernel, step_s):
    """The VALU-issue view of the same kernel (its own analysis says issue-bound), from the committed
    rocprofv3 PMC passes (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU; profiles/) over this run's time per step.  The headline is
    `valu_busy_frac`: cycles in which a SIMD's VALU was issuing / the cycles of a step (SQ_ACTIVE_INST_VALU counts
    quad-cycles summed over the device's 1024 SIMDs).  `frac` prices the instruction count against the f32 issue peak of
    MI355X_MICROARCH.md: a wave64 `v_fma_f32` occupies its SIMD-32 for 2 cycles (float64: 4), i.e. 256 CUs x 4 SIMDs x
    2.4 GHz / 2 = 1.23e12 wave-instructions/s."""
    for d, src in _profile_records("valu"):
        if d.get("envs") != E or d.get("agents") != N or d.get("kernel", "").split("(")[0] not in kernel:
            continue
        per_step = d.get("steps_per_launch", 1)
        insts = d["valu_insts_per_launch"] / per_step
        peak = 256 * 4 * 2.4e9 / 2.0
        out = {"valu_insts_per_step": insts, "valu_insts_per_agent_step": insts / (E * N),
               "achieved_insts_per_s": insts / step_s, "peak_insts_per_s": peak, "frac": insts / step_s / peak,
               "source": src,
               "note": "peak = f32 rate (2 cycles per wave64 instruction on a SIMD-32; float64 instructions take 4, "
                       "transcendentals more): frac is a lower bound of how busy the issue pipes are -- valu_busy_frac "
                       "(SQ_ACTIVE_INST_VALU) measures it"}
        if d.get("valu_busy_cycles_per_simd"):
            busy = d["valu_busy_cycles_per_simd"] / per_step
            out["valu_busy_frac"] = busy / (step_s * 2.4e9)
            out["busy_cycles_per_valu_inst"] = busy * 1024.0 / insts
        return out
    return None


def respawn(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, exactly the
    command the driver's launcher form uses) and hand their exit code on; rank 0 prints the one JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["CAGPU_BENCH_SPAWNED"] = "1"
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    for line in r.stdout.decode(errors="replace").splitlines():   # stdout carries the ONE JSON line; whatever the
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")   # backends print goes to stderr
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--workload", choices=["rvo10", "ga3c20", "crowd50_laser"], default="rvo10")
    ap.add_argument("--mode", choices=["lookahead", "step", "graph", "rollout"], default=None,
                    help="lookahead (default for the RVO workload): env.step(None) served from a ring of L steps computed "
                         "ahead in one launch of the fused n-step kernel, every step's outputs kept (the product path of "
                         "CollisionAvoidanceEnv.step(None)); step: one launch per env.step, submitted call by call (what a "
                         "caller with external actions gets; the default of the ga3c20 / crowd50_laser workloads, which need "
                         "a second kernel between two steps); graph: the same K launches captured once into a HIP graph and "
                         "replayed; rollout: all K steps fused in one launch, only the last step's outputs kept")
    ap.add_argument("--lookahead", type=int, default=0,
                    help="ring length L of --mode lookahead; 0: the largest divisor of --steps that is <= 64 (a timed block "
                         "is then exactly steps / L launches)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: initialise torch.distributed (--backend, default nccl = RCCL) with world_size 1 anyway and "
                         "run every collective of the N > 1 path (barriers, the MAX all-reduce of the block table, the "
                         "episode-statistics all-reduce, the all-gather of per-rank times)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ga3c-fused", action="store_true",
                    help="ga3c20: cagpu_ga3c computes the observation rows it needs from the state itself (obs = NULL: sensing + "
                         "inference fused in one kernel) instead of reading the rows the step kernel stored")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="A/B: without CaState.next_action, i.e. the unpipelined step kernel (policy query at the start of the step)")
    ap.add_argument("--min-warm-seconds", type=float, default=0.3,
                    help="untimed steady-state warm-up on top of --warmup (launches until this much time has passed)")
    ap.add_argument("--min-timed-seconds", type=float, default=0.5,
                    help="the timed block of EXACTLY --steps steps is repeated until this much device time has been measured "
                         "(at most --max-blocks blocks); the line reports the MEDIAN block.  0: a single block")
    ap.add_argument("--max-blocks", type=int, default=1000)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the rollout / two-stream extras (profiling runs: only the headline kernel is launched)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI)")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (lets a 1-GPU box exercise the N>1 code path with gloo)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(a)   # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to report a line whose n_gpus "
                 "is not the number of ranks that ran" % (a.gpus, world))
    if world > 1 and not a.share_device and torch.cuda.device_count() < world:
        sys.exit("bench.py: --gpus %d but only %d device(s) visible (one rank per GPU; --share-device is for 1-GPU tests)"
                 % (world, torch.cuda.device_count()))
    if a.mode is None:
        a.mode = "lookahead" if a.workload == "rvo10" else "step"
    if a.mode == "lookahead" and a.workload != "rvo10":
        sys.exit("bench.py: --mode lookahead needs a workload without work between two steps (rvo10)")
    dist_on = world > 1 or a.force_dist     # every collective below runs whenever a process group exists
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:   # --force-dist without a launcher: a one-rank group on a free port
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            sk.close()
        if a.share_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from gym_collision_avoidance_amd import _native as nat
    from gym_collision_avoidance_amd import core
    from gym_collision_avoidance_amd.sharding import shard_env_ids, reduce_episode_stats

    E = a.envs
    sim, table, N, K = build_workload(a.workload, E, dev, rank, world, pipeline=not a.no_pipeline, agents=a.agents)
    sim.ga3c_fused = bool(a.ga3c_fused)
    off, stride = shard_env_ids(rank, world, E)

    graph = {}
    L = 1
    if a.mode == "lookahead":
        L = a.lookahead if a.lookahead > 0 else max(d for d in range(1, 65) if a.steps % d == 0)
        if a.steps % L:
            sys.exit("bench.py: --lookahead %d does not divide --steps %d (a timed block must be whole launches)" % (L, a.steps))
        sim.enable_lookahead(L, fresh=True)

    def run(n):
        if a.mode == "lookahead":
            for _ in range(n):
                sim.step_lookahead()
        elif a.mode == "graph" and n in graph:
            graph[n].replay()
        elif a.mode == "rollout":
            sim.rollout(n)
        elif a.workload == "crowd50_laser":
            for _ in range(n):
                sim.step()
                sim.laserscan()
        else:
            for _ in range(n):
                sim.step()

    # ---- untimed: one pass through the statistics reduction (its first call loads the reduce kernel's code object
    # and, for N > 1, builds the RCCL communicator) and through the timing events (torch creates them lazily), the
    # caller's W warm-up steps, then launches until the device has been busy for >= 0.3 s (clocks and caches in steady
    # state whatever W was).  The warm-up ends directly at the synchronize that opens the timed region: no idle gap in
    # which the device could drop its clocks.
    import gc
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    reduce_episode_stats(sim.episode_stats(), world, force=dist_on)
    chunk = 50 if a.mode in ("step", "graph") else (a.steps if a.mode == "rollout" else L * max(1, 50 // L))
    run(a.warmup if a.mode != "lookahead" else -(-a.warmup // L) * L)   # (whole rings: a timed block starts at a ring boundary)
    torch.cuda.synchronize(dev)
    if a.mode == "graph":  # capture the launches of K steps (and of the warm-up chunk) once; run() then replays them
        for n in sorted({a.steps, 50}):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    sim.step() if a.workload != "crowd50_laser" else (sim.step(), sim.laserscan())
            graph[n] = g
        torch.cuda.synchronize(dev)
    gc.collect()
    gc.disable()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < a.min_warm_seconds:
        run(chunk)
        torch.cuda.synchronize(dev)

    # ---- timed: a BLOCK is EXACTLY a.steps steps between barrier + synchronize on both sides, nothing else inside.  One
    # block of the driver's 20 steps is 0.3 ms of device time -- one preempted launch moves it by 5 %, and a utilisation
    # sampler never sees it -- so the block is repeated back to back until --min-timed-seconds of device time have been
    # measured and the line reports the MEDIAN block (per block: max over ranks); first / min / max are reported beside it.
    def timed_block():
        if dist_on:
            dist.barrier()
            run(20 if a.mode == "step" else (50 if a.mode == "graph" else (L if a.mode == "lookahead" else a.steps)))   # the barrier idled the device: bring it back before the clock starts
        torch.cuda.synchronize(dev)
        ev0.record()            # same stream the kernels are launched on (torch's current stream)
        t0 = time.perf_counter()
        run(a.steps)
        ev1.record()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, ev0.elapsed_time(ev1)

    blocks = [timed_block()]
    n_blocks = 1
    if a.min_timed_seconds > 0:
        n_blocks = int(min(max(1, a.max_blocks if world == 1 else min(a.max_blocks, 200)),
                           max(1, np.ceil(a.min_timed_seconds / max(blocks[0][1] * 1e-3, 1e-6)))))
    if dist_on:   # every rank must run the same number of blocks (there is a barrier in each)
        nb = torch.tensor([n_blocks], dtype=torch.int64, device=dev)
        dist.broadcast(nb, 0)
        n_blocks = int(nb.item())
    for _ in range(n_blocks - 1):
        blocks.append(timed_block())
    gc.enable()
    if dist_on:
        dist.barrier()
    bt = torch.tensor(blocks, dtype=torch.float64, device=dev)      # [blocks, (wall s, events ms)]
    bt_local = bt.clone()
    if dist_on:
        dist.all_reduce(bt, op=dist.ReduceOp.MAX)
    walls, gpus = bt[:, 0].cpu().numpy(), bt[:, 1].cpu().numpy()
    mid = int(np.argsort(walls)[len(walls) // 2])                    # the median block (by wall clock)
    wall, gpu_ms = float(walls[mid]), float(gpus[mid])
    gpu_ms_local = float(bt_local[mid, 1].item())
    kernel_name = nat.lib().cagpu_last_kernel().decode()       # (of the timed launches: episode_stats() below may rewind the ring)
    stats_local = sim.episode_stats()
    stats = reduce_episode_stats(stats_local, world, force=dist_on)   # the only collective (8 counters), outside the clock
    torch.cuda.synchronize(dev)
    # ---- multi-GPU evidence the driver can read from the JSON line: ranks seen, per-rank device time of the SAME K
    # steps, and the latency of the one collective (the 8-counter all-reduce), measured outside the step clock
    per_rank_ms, allreduce_us = [gpu_ms_local / a.steps], None
    ranks_seen = 1
    per_rank_stats = None
    if dist_on:
        mine = torch.tensor([gpu_ms_local / a.steps], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(x.item()) for x in every]
        from gym_collision_avoidance_amd.sharding import gather_episode_stats
        per_rank_stats = [[float(x) for x in row] for row in gather_episode_stats(stats_local).cpu().numpy()]
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one)                # counts the ranks that actually took part
        ranks_seen = int(round(float(one.item())))
        probe = sim.episode_stats()
        for _ in range(5):
            reduce_episode_stats(probe, world, force=True)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t_c = time.perf_counter()
        for _ in range(50):
            reduce_episode_stats(probe, world, force=True)
        torch.cuda.synchronize(dev)
        allreduce_us = (time.perf_counter() - t_c) / 50 * 1e6
        if ranks_seen != a.gpus:
            sys.exit("bench.py: %d ranks took part, --gpus %d asked for" % (ranks_seen, a.gpus))
        if world == 1 and not torch.equal(stats, stats_local):
            sys.exit("bench.py: the one-rank all-reduce changed the episode statistics")

    if rank == 0:
        agent_steps = float(world) * E * N * a.steps
        value = agent_steps / wall
        launches = 1 if a.mode == "rollout" else a.steps // L
        steps_per_launch = a.steps // launches
        bytes_per_launch = algorithmic_bytes_per_agent_step(K) * E * N * steps_per_launch
        kern_s = gpu_ms * 1e-3 / launches          # average launch duration (HIP events over the timed region)
        achieved_ev = bytes_per_launch / kern_s / 1e9
        # `value` is whole-job throughput on the WALL clock of the timed block, so the headline fraction is priced on the same
        # clock (what the driver's own timing reproduces); the HIP-event figure (device time of the same launches) is beside it
        achieved = bytes_per_launch * launches / wall / 1e9
        traffic, traffic_src = (measured_traffic_bytes(E, N, kernel_name, steps_per_launch)
                                if a.mode in ("step", "lookahead") else (None, None))
        total_envs = world * E
        shape = {(4096, 10): "the metric's size (configs[3] = 8 of these shards)", (1024, 10): "BASELINE configs[1]",
                 (32768, 10): "BASELINE configs[3] as ONE batch on one GPU"}.get((E, N), "a side configuration")
        out = {
            "metric": "agent-steps/sec at %d envs x %d agents (RVO)" % (E, N), "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall * 1e3 / a.steps,
            "event_ms_per_step": gpu_ms / a.steps,   # HIP events around the same K steps (device time only)
            "timed_blocks": {"blocks": len(walls), "steps_per_block": a.steps, "reported": "median block by wall clock",
                             "ms_per_step_first": float(walls[0]) * 1e3 / a.steps, "ms_per_step_min": float(walls.min()) * 1e3 / a.steps,
                             "ms_per_step_max": float(walls.max()) * 1e3 / a.steps,
                             "device_seconds_timed": float(gpus.sum()) * 1e-3},
            "suspect": bool(abs(wall * 1e3 - gpu_ms) > 0.2 * gpu_ms),  # host wall clock and device time disagree by > 20 %
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs/GPU x %d agents (%s; %d envs in all), RVOPolicy(ORCA) + UnicycleDynamics + "
                                   "OtherAgentsStatesSensor K=%d closest_first, DT=0.1, fixture %d_agents_500_cases, auto-reset; "
                                   "every step's obs / rewards / done / game_over handed out" % (E, N, shape, total_envs, K, N),
                       "envs_per_gpu": E, "agents": N, "parallelism": "env-shard x%d" % world,
                       "launch_mode": ("lookahead-%d" % L) if a.mode == "lookahead" else a.mode,
                       "launch_mode_note": {
                           "lookahead": "env.step(None) served from a ring of %d steps computed ahead by ONE launch of the fused "
                                        "n-step kernel (cagpu_rollout_ring); bit-identical to one launch per step "
                                        "(tests/test_gpu_ring.py); a timed block of %d steps = %d launches + %d state snapshots"
                                        % (L, a.steps, launches, launches),
                           "step": "one cagpu_step launch per env.step",
                           "graph": "one cagpu_step launch per env.step, replayed from a HIP graph",
                           "rollout": "cagpu_rollout: all steps in one launch, only the LAST step's outputs are kept"}[a.mode]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "clock": "wall clock of the timed block (the clock of `value`); *_events: HIP events around the same launches",
                         "achieved_events": achieved_ev, "frac_events": achieved_ev / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_unit": ("bytes per launch: rocprofv3 FETCH_SIZE + WRITE_SIZE passes over this kernel (calibrated on a copy of "
                                          "known size), committed as %s; not re-measured in this run" % traffic_src) if traffic is not None else
                                         "no committed counter pass for this geometry / kernel",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel": kernel_name, "avg_launch_us": kern_s * 1e6, "steps_per_launch": steps_per_launch,
                         "us_per_step": kern_s * 1e6 / steps_per_launch,
                         "algorithmic_bytes_per_agent_step": algorithmic_bytes_per_agent_step(K),
                         "valu": valu_block(E, N, kernel_name, kern_s / steps_per_launch) if a.workload == "rvo10" else None},
            "episode_stats": dict(zip(core.STAT_NAMES, [float(x) for x in stats.cpu().numpy()])),
            "ranks_seen": ranks_seen,
            "distributed": ({"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist_on else None),
            "per_rank_event_ms_per_step": per_rank_ms,
            "stats_allreduce_us": allreduce_us,   # the ONLY collective (RCCL all-reduce of 8 float64 counters), off the step path
            "episode_stats_per_rank": per_rank_stats,   # (N > 1 / --force-dist: the all-gathered shard counters; their sum is episode_stats)
            "provenance": provenance(),
        }
        if a.workload != "rvo10":
            extra_workload(out, a, sim, core, E, N, K, dev, torch)
        extras = world == 1 and a.mode in ("step", "lookahead") and a.workload == "rvo10" and not a.no_extras
        if extras and a.mode == "lookahead":
            # beside the headline: the same workload one cagpu_step launch per step (what a caller with external actions gets)
            n1 = max(a.steps, 500)
            sim.sync()
            for _ in range(50):
                sim.step()
            s1 = _time_launches(lambda: sim.step(), n1, torch, dev)
            out["single_launch"] = {"value": E * N / s1, "unit": "agent-steps/s", "us_per_step": s1 * 1e6, "launches": n1,
                                    "kernel": nat.lib().cagpu_last_kernel().decode(),
                                    "roofline_frac": algorithmic_bytes_per_agent_step(K) * E * N / s1 / 1e9 / HBM_PEAK_GBS,
                                    "note": "one cagpu_step launch per env.step(None) (HIP events over %d back-to-back launches): a "
                                            "launch ends with its slowest workgroup, the fused n-step kernel of the headline runs "
                                            "at the mean" % n1}
        if extras:
            # extra: the same K steps fused into ONE cagpu_rollout launch (env_utils.run_episode's loop on the device)
            torch.cuda.synchronize(dev)
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record()
            sim.rollout(a.steps)
            r1.record()
            torch.cuda.synchronize(dev)
            rms = r0.elapsed_time(r1)
            out["rollout"] = {"value": E * N * a.steps / (rms * 1e-3), "unit": "agent-steps/s",
                              "ms_per_step": rms / a.steps, "launches": 1,
                              "note": "same workload, %d steps in one launch (state stays in registers/LDS between "
                                      "steps; observations, rewards and done flags are still written every step)" % a.steps}
        if extras and a.mode == "step" and E % 2 == 0:
            # extra: the same batch as two half-batches on two HIP streams (envs are independent): the tail of one
            # launch -- a launch ends with its slowest workgroup -- overlaps with the body of the other
            halves, streams = [], [torch.cuda.Stream(device=dev) for _ in range(2)]
            for h in range(2):
                sh = core.BatchedSim(core.make_params(E // 2, N, max_obs=K), device=dev)
                sh.set_plugins(nat.POL_RVO, nat.DYN_UNICYCLE)
                sh.set_fixture_table(table, env_id_offset=off + h * (E // 2), case_stride=stride)
                sh.reset_from_table()
                halves.append(sh)

            # the launches of both chains are captured into ONE HIP graph (two parallel branches) and replayed: submitted
            # call by call from Python, two launches per step are host-bound (~11 us each)
            nrep = min(a.steps, 500)
            for sh in halves:
                for _ in range(max(a.warmup, 5)):
                    sh.step()
            torch.cuda.synchronize(dev)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                cur = torch.cuda.current_stream(dev)
                for st_ in streams:
                    st_.wait_stream(cur)
                for sh, st_ in zip(halves, streams):
                    with torch.cuda.stream(st_):
                        for _ in range(nrep):
                            sh.step()
                for st_ in streams:
                    cur.wait_stream(st_)
            g2.replay()
            torch.cuda.synchronize(dev)
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(1, a.steps // nrep)
            q0.record()
            for _ in range(reps):
                g2.replay()
            q1.record()
            torch.cuda.synchronize(dev)
            t2 = q0.elapsed_time(q1) * 1e-3
            out["two_streams"] = {"value": E * N * nrep * reps / t2, "unit": "agent-steps/s",
                                  "ms_per_step": t2 * 1e3 / (nrep * reps), "launches_per_step": 2,
                                  "note": "same workload as 2 x %d envs: two chains of %d single-step launches each, as the "
                                          "two branches of one HIP graph (the ramp-up and the tail of one chain's launch "
                                          "overlap with the body of the other's); not the headline: per-launch durations "
                                          "overlap, so the roofline above is quoted for the single-chain launch" % (E // 2, nrep)}
        if extras:
            out["env_api"] = env_api_rates(E, N, min(max(a.steps, 640), 1280), torch, dev)
        if world == 1 and not a.no_cpu_baseline and a.workload == "rvo10":
            out["cpu_baseline"] = cpu_baseline(N, K)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
