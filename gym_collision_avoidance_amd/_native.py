"""ctypes binding of libcagpu.so (include/cagpu.h).  The product path: there is NO CPU fallback --
if the HIP library is missing or fails to load this module raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CAGPU_LIB: load another build of the same ABI (the -DCAGPU_ABLATE experiment build of scratch/); the library itself
# reads no environment variable
LIB_PATH = os.environ.get("CAGPU_LIB") or os.path.join(HERE, "libcagpu.so")

# ---- constants mirrored from include/cagpu.h
CA_OK, CA_EINVAL, CA_EUNSUPPORTED, CA_ELAUNCH, CA_ENODEVICE = 0, -1, -2, -3, -4
AT_GOAL, WAS_AT_GOAL, IN_COLLISION, WAS_IN_COLLISION, OUT_OF_TIME, DONE, IS_LEARNING, STILL_LEARNING = (
    1 << 0, 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5, 1 << 6, 1 << 7)
POLICY_SHIFT, DYNAMICS_SHIFT = 8, 12
ABSENT, PLAN_VALID = 1 << 16, 1 << 17
ABI_VERSION = 11  # CAGPU_VERSION of include/cagpu.h: the struct layouts below mirror THAT header
POL_RVO, POL_NONCOOP, POL_STATIC, POL_EXTERNAL, POL_LEARNING, POL_LEARNING_GA3C, POL_GA3C_CADRL = range(7)
DYN_UNICYCLE, DYN_MAX_TURN_RATE, DYN_EXTERNAL = range(3)
SORT_CLOSEST_FIRST, SORT_CLOSEST_LAST, SORT_TIME_TO_IMPACT = range(3)
OVER_ALL_DONE, OVER_AGENT0, OVER_LEARNING_DONE = range(3)

_P = C.c_void_p


class CaParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_envs", "num_agents", "max_obs", "sort_mode", "game_over_mode",
                                         "rvo_max_neighbors", "obs_clip", "ragged")] + \
               [(n, C.c_double) for n in ("dt", "near_goal_threshold", "max_time_ratio", "getting_close_range",
                                          "sensing_horizon", "reward_at_goal", "reward_collision", "reward_time_step",
                                          "reward_wiggly", "wiggly_threshold", "reward_min", "reward_max",
                                          "rvo_time_horizon", "rvo_collab_coeff", "max_heading_change",
                                          "reward_collision_wall", "rvo_dt")]


STATE_FIELDS = ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "radius", "pref_speed",
                "time_remaining", "t", "slt", "ep_reward", "last_action", "flags", "step_num", "episode_step",
                "reset_count", "env_stats", "next_action", "turning_dir", "rvo_collab", "rvo_heading_noise", "ext_state")
OUT_FIELDS = ("obs", "rewards", "done", "game_over", "actions", "orca_vel", "workspace")


class CaState(C.Structure):
    _fields_ = [(n, _P) for n in STATE_FIELDS]


class CaOut(C.Structure):
    _fields_ = [(n, _P) for n in OUT_FIELDS] + [("workspace_bytes", C.c_uint64)]


class CaAutoReset(C.Structure):
    _fields_ = [("table", _P), ("n_cases", C.c_int32), ("env_id_offset", C.c_int64), ("case_stride", C.c_int64),
                ("reset_obs", _P), ("reset_plan", _P), ("heading_seed", C.c_uint64)]


class CaMap(C.Structure):
    _fields_ = [("static_bits", _P), ("rows", C.c_int32), ("cols", C.c_int32), ("cell", C.c_double),
                ("origin_r", C.c_double), ("origin_c", C.c_double)]


class CaScan(C.Structure):
    _fields_ = [("hist", _P), ("out", _P), ("num_beams", C.c_int32), ("num_to_store", C.c_int32),
                ("num_ranges", C.c_int32), ("reserved0", C.c_int32), ("min_angle", C.c_double),
                ("max_angle", C.c_double), ("range_res", C.c_double), ("max_range", C.c_double)]


NET_FIELDS = ("lstm_kernel", "lstm_bias", "layer1_kernel", "layer1_bias", "layer2_kernel", "layer2_bias", "fc1_kernel",
              "fc1_bias", "logits_kernel", "logits_bias", "input_mean", "input_std")


class CaNet(C.Structure):
    _fields_ = [(n, _P) for n in NET_FIELDS] + [("rows_scratch", _P), ("agent_net", _P), ("net_index", C.c_int32),
                                                ("reserved0", C.c_int32), ("packed", _P)]


EXPORTS = ("cagpu_version", "cagpu_last_error", "cagpu_last_kernel", "cagpu_reset", "cagpu_step", "cagpu_step_map", "cagpu_rollout",
           "cagpu_orca", "cagpu_observe", "cagpu_laserscan", "cagpu_ga3c", "cagpu_generate_cases", "cagpu_generate_cases_ragged", "cagpu_plan", "cagpu_debug_libm", "cagpu_device_faults", "cagpu_workspace_bytes",
           "cagpu_ga3c_packed_bytes", "cagpu_ga3c_pack", "cagpu_rollout_ring", "cagpu_ring_snapshots", "cagpu_debug_copy8", "cagpu_device_faults_async")

_lib = None


class CagpuError(RuntimeError):
    pass


def lib():
    """Load libcagpu.so; raise loudly if it is not there (no eager / CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CagpuError("libcagpu.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950); there is no CPU fallback for the hot path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.cagpu_version.restype = C.c_int
    L.cagpu_last_error.restype = C.c_char_p
    got = L.cagpu_version()
    if got != ABI_VERSION:  # a stale or experiment build would silently mis-read the structs above
        raise CagpuError("%s reports ABI version %d, this binding mirrors version %d of include/cagpu.h -- rebuild "
                         "(python -m gym_collision_avoidance_amd.build_native)" % (LIB_PATH, got, ABI_VERSION))
    PP, PS, PO, PA = C.POINTER(CaParams), C.POINTER(CaState), C.POINTER(CaOut), C.POINTER(CaAutoReset)
    L.cagpu_reset.argtypes = [PP, PS, PO, _P, _P, _P, _P]
    L.cagpu_step.argtypes = [PP, PS, PO, _P, PA, _P]
    L.cagpu_rollout.argtypes = [PP, PS, PO, _P, PA, C.c_int32, _P]
    L.cagpu_rollout_ring.argtypes = [PP, PS, PO, _P, PA, C.c_int32, C.c_int64, _P]
    L.cagpu_ring_snapshots.argtypes = [PP, PS, PO, PA, C.c_int32]
    L.cagpu_debug_copy8.argtypes = [C.c_int64, _P, _P, _P]
    L.cagpu_step_map.argtypes = [PP, PS, PO, _P, PA, C.POINTER(CaMap), _P]
    L.cagpu_laserscan.argtypes = [PP, PS, C.POINTER(CaMap), C.POINTER(CaScan), _P]
    L.cagpu_observe.argtypes = [PP, PS, PO, _P]
    L.cagpu_plan.argtypes = [PP, PS, _P]
    L.cagpu_orca.argtypes = [C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int32,
                             C.c_float, _P, _P]
    L.cagpu_ga3c.argtypes = [PP, PS, _P, C.POINTER(CaNet), _P, _P, _P]
    L.cagpu_generate_cases.argtypes = [C.c_int64, C.c_int32] + [C.c_double] * 6 + [C.c_uint64, _P, _P, _P]
    L.cagpu_generate_cases_ragged.argtypes = ([C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32] + [C.c_double] * 4 +
                                              [C.c_uint64, _P, _P, _P, _P])
    L.cagpu_device_faults.argtypes = [_P, C.c_int32]
    L.cagpu_device_faults_async.argtypes = [_P, _P]
    L.cagpu_workspace_bytes.argtypes = [PP]
    L.cagpu_workspace_bytes.restype = C.c_uint64
    L.cagpu_ga3c_packed_bytes.argtypes = []
    L.cagpu_ga3c_packed_bytes.restype = C.c_uint64
    L.cagpu_ga3c_pack.argtypes = [C.POINTER(CaNet), _P, C.c_uint64, _P]
    L.cagpu_debug_libm.argtypes = [C.c_int32, C.c_int32, _P, _P, _P, _P]
    for n in EXPORTS:
        getattr(L, n)  # AttributeError if a declared symbol is missing
        if n not in ("cagpu_last_error", "cagpu_last_kernel", "cagpu_workspace_bytes", "cagpu_ga3c_packed_bytes"):
            getattr(L, n).restype = C.c_int
    L.cagpu_last_error.restype = C.c_char_p
    L.cagpu_last_kernel.restype = C.c_char_p
    _lib = L
    return L


def device_faults(clear=True):
    """cagpu_device_faults: the current device's fault word (0 in normal operation); synchronises the device."""
    v = C.c_uint32(0)
    check(lib().cagpu_device_faults(C.byref(v), 1 if clear else 0))
    return int(v.value)


def debug_libm(op, a, b=None):
    """cagpu_debug_libm on numpy float64 arrays -> (out0, out1): the device's own atan2 / heading sincos / lean divide and
    square root (parity hook, see include/cagpu.h)."""
    import numpy as np
    a = np.ascontiguousarray(a, np.float64).reshape(-1)
    b = None if b is None else np.ascontiguousarray(b, np.float64).reshape(-1)
    o0, o1 = np.empty_like(a), np.empty_like(a)
    check(lib().cagpu_debug_libm(op, a.size, a.ctypes.data, None if b is None else b.ctypes.data, o0.ctypes.data, o1.ctypes.data))
    return o0, o1


def check(rc):
    if rc != 0:
        raise CagpuError("cagpu error %d: %s" % (rc, lib().cagpu_last_error().decode()))
