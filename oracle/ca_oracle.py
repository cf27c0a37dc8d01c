"""oracle/ca_oracle.py -- TEST INFRASTRUCTURE.  ctypes front-end for oracle/_build/libca_oracle.so (the CPU
restatement of the reference step, oracle/ca_oracle.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module -- as the checker / reported CPU baseline, never as the product path.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libca_oracle.so")

# flag bits / ids: mirror oracle/ca_oracle.h
AT_GOAL, WAS_AT_GOAL, IN_COLLISION, WAS_IN_COLLISION, OUT_OF_TIME, DONE, IS_LEARNING, STILL_LEARNING = (
    1 << 0, 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5, 1 << 6, 1 << 7)
ABSENT = 1 << 16
POL_RVO, POL_NONCOOP, POL_STATIC, POL_EXTERNAL, POL_LEARNING, POL_LEARNING_GA3C, POL_GA3C_CADRL = range(7)
DYN_UNICYCLE, DYN_MAX_TURN_RATE, DYN_EXTERNAL = range(3)
SORT_CLOSEST_FIRST, SORT_CLOSEST_LAST, SORT_TIME_TO_IMPACT = range(3)
OVER_ALL_DONE, OVER_AGENT0, OVER_LEARNING_DONE = range(3)

_D = C.POINTER(C.c_double)
_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_I32 = C.POINTER(C.c_int32)
_U8 = C.POINTER(C.c_uint8)


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_envs", "num_agents", "max_obs", "sort_mode", "game_over_mode",
                                         "rvo_max_neighbors", "obs_clip", "ragged")] + \
               [(n, C.c_double) for n in ("dt", "near_goal_threshold", "max_time_ratio", "getting_close_range",
                                          "sensing_horizon", "reward_at_goal", "reward_collision", "reward_time_step",
                                          "reward_wiggly", "wiggly_threshold", "reward_min", "reward_max",
                                          "rvo_time_horizon", "rvo_collab_coeff", "max_heading_change",
                                          "reward_collision_wall", "rvo_dt")]


_STATE_F64 = ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "radius", "pref_speed",
              "time_remaining", "t", "slt", "ep_reward", "turning_dir")


class OrcState(C.Structure):
    _fields_ = [(n, _D) for n in _STATE_F64] + [("last_action", _F), ("flags", _U32), ("policy", _I32),
                                                ("dynamics", _I32), ("step_num", _I32), ("episode_step", _I32),
                                                ("reset_count", _I32), ("env_stats", _D), ("rvo_collab", _F),
                                                ("rvo_heading_noise", _D), ("ext_state", _D)]


class OrcOut(C.Structure):
    _fields_ = [("obs", _D), ("rewards", _D), ("done", _U8), ("game_over", _U8), ("actions", _F), ("orca_vel", _F)]


class OrcMap(C.Structure):
    _fields_ = [("static_map", _U8), ("rows", C.c_int32), ("cols", C.c_int32), ("cell", C.c_double),
                ("origin_r", C.c_double), ("origin_c", C.c_double)]


class OrcScan(C.Structure):
    _fields_ = [("hist", _U8), ("out", _D), ("num_beams", C.c_int32), ("num_to_store", C.c_int32),
                ("num_ranges", C.c_int32), ("min_angle", C.c_double), ("max_angle", C.c_double),
                ("range_res", C.c_double), ("max_range", C.c_double)]


_lib = None


def build():
    subprocess.check_call(["make", "-C", HERE, "-s", "_build/libca_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.ca_oracle_round2.restype = C.c_double
        _lib.ca_oracle_round2.argtypes = [C.c_double]
    return _lib


def default_params(num_envs, num_agents, max_obs=None, dt=0.1, max_time_ratio=8.0, sort_mode=SORT_CLOSEST_FIRST,
                   game_over_mode=OVER_ALL_DONE, rvo_max_neighbors=None, near_goal=0.2, getting_close=0.2,
                   obs_clip=None, ragged=0):
    """Constants of the reference Config (config.py:28-86) for an EvaluateConfig-style run (config.py:193-200)."""
    p = OrcParams()
    p.num_envs, p.num_agents = num_envs, num_agents
    p.max_obs = num_agents - 1 if max_obs is None else max_obs
    p.obs_clip = p.max_obs if obs_clip is None else obs_clip
    p.ragged = int(ragged)
    p.sort_mode, p.game_over_mode = sort_mode, game_over_mode
    p.rvo_max_neighbors = num_agents if rvo_max_neighbors is None else rvo_max_neighbors
    p.dt, p.near_goal_threshold, p.max_time_ratio = dt, near_goal, max_time_ratio
    p.getting_close_range, p.sensing_horizon = getting_close, math.inf
    p.reward_at_goal, p.reward_collision, p.reward_time_step = 1.0, -0.25, 0.0
    p.reward_wiggly, p.wiggly_threshold = 0.0, math.inf
    p.reward_min, p.reward_max = -0.25, 1.0
    p.rvo_time_horizon, p.rvo_collab_coeff = 5.0, 0.5
    p.max_heading_change = math.pi / 3
    p.reward_collision_wall, p.rvo_dt = -0.25, dt
    return p


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle(object):
    """Numpy-owned SoA state + the oracle entry points."""

    def __init__(self, params):
        self.p = params
        E, N, K = params.num_envs, params.num_agents, params.max_obs
        self.E, self.N, self.K, self.W = E, N, K, 6 + 7 * K
        self.s = {n: np.zeros(E * N, np.float64) for n in _STATE_F64}
        self.s["last_action"] = np.zeros((E * N, 2), np.float32)
        self.s["flags"] = np.zeros(E * N, np.uint32)
        self.s["policy"] = np.zeros(E * N, np.int32)
        self.s["dynamics"] = np.zeros(E * N, np.int32)
        self.s["step_num"] = np.zeros(E * N, np.int32)
        self.s["episode_step"] = np.zeros(E, np.int32)
        self.s["reset_count"] = np.zeros(E, np.int32)
        self.s["env_stats"] = np.zeros((E, 8), np.float64)
        self.obs = np.zeros((E, N, self.W), np.float64)
        self.rewards = np.zeros((E, N), np.float64)
        self.done = np.zeros((E, N), np.uint8)
        self.game_over = np.zeros(E, np.uint8)
        self.actions = np.zeros((E, N, 2), np.float32)
        self.orca_vel = np.zeros((E, N, 2), np.float32)
        self.cmap = None
        self.net = None           # oracle/ga3c_ref.GA3CNet, created on first use
        self.ga3c_index = None    # [E*N] last action indices chosen by the network (-1 = not queried)
        self._bind()

    def _bind(self):
        st = OrcState()
        for n, t in OrcState._fields_:
            if n in self.s:
                setattr(st, n, _ptr(self.s[n], t))
        self.cs = st
        self.co = OrcOut(_ptr(self.obs, _D), _ptr(self.rewards, _D), _ptr(self.done, _U8), _ptr(self.game_over, _U8),
                         _ptr(self.actions, _F), _ptr(self.orca_vel, _F))

    def set_rvo_stochastic(self, collab=None, heading_noise=None):
        """this step's draws of RVOPolicy's stochastic branches (RVOPolicy.py:77-90, :118-119): float32 [E*N] ego
        collaboration coefficients and / or float64 [E*N] heading noise; None switches a branch off"""
        for name, arr, dt in (("rvo_collab", collab, np.float32), ("rvo_heading_noise", heading_noise, np.float64)):
            if arr is None:
                self.s.pop(name, None)
            else:
                self.s[name] = np.ascontiguousarray(np.asarray(arr, dt).reshape(-1))
        self._bind()

    def set_ext_state(self, ext_state=None):
        """float64 [E*N, 5] = px, py, vx, vy, heading taken at the move by DYN_EXTERNAL agents (NaN rows: none), or None"""
        if ext_state is None:
            self.s.pop("ext_state", None)
        else:
            self.s["ext_state"] = np.ascontiguousarray(np.asarray(ext_state, np.float64).reshape(-1, 5))
        self._bind()

    def set_policies(self, policy, dynamics=None):
        pol = np.broadcast_to(np.asarray(policy, np.int32).reshape(-1, self.N) if np.ndim(policy) else policy,
                              (self.E, self.N)).reshape(-1)
        self.s["policy"][:] = pol
        if dynamics is not None:
            self.s["dynamics"][:] = np.broadcast_to(np.asarray(dynamics, np.int32), (self.E, self.N)).reshape(-1)
        learn = (self.s["policy"] == POL_LEARNING) | (self.s["policy"] == POL_LEARNING_GA3C)
        self.s["flags"][:] = np.where(learn, IS_LEARNING | STILL_LEARNING, 0).astype(np.uint32)

    def reset(self, cases, headings=None, mask=None):
        cases = np.ascontiguousarray(cases, np.float64).reshape(self.E, self.N, 6)
        h = None if headings is None else np.ascontiguousarray(headings, np.float64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = lib().ca_oracle_reset(C.byref(self.p), C.byref(self.cs), C.byref(self.co), _ptr(cases, _D),
                                   None if h is None else _ptr(h, _D), None if m is None else _ptr(m, _U8))
        assert rc == 0
        return self.obs

    def ga3c_query(self, ext_actions=None):
        """GA3CCADRLPolicy.find_next_action for every live GA3C-CADRL agent on the current observation (float32, like
        the TF feed): the action index goes into ext[..., 0] (consumed by the C++ step like LEARNING_GA3C)."""
        from .ga3c_ref import GA3CNet
        if self.net is None:
            self.net = GA3CNet()
        e = np.zeros((self.E, self.N, 2), np.float64) if ext_actions is None else \
            np.array(ext_actions, dtype=np.float64).reshape(self.E, self.N, 2)
        live = (self.s["policy"] == POL_GA3C_CADRL) & ((self.s["flags"] & DONE) == 0)
        self.ga3c_index = np.full(self.E * self.N, -1, np.int64)
        if live.any():
            rows = self.obs.reshape(-1, self.W)[live].astype(np.float32)
            self.ga3c_index[live] = self.net.action_index(rows)
            e.reshape(-1, 2)[live, 0] = self.ga3c_index[live]
            e.reshape(-1, 2)[live, 1] = 0.0
        return e

    def step(self, ext_actions=None):
        if (self.s["policy"] == POL_GA3C_CADRL).any():
            ext_actions = self.ga3c_query(ext_actions)
        e = None if ext_actions is None else np.ascontiguousarray(ext_actions, np.float64)
        if self.cmap is not None:  # env.py:494-506: wall collisions against the static map
            rc = lib().ca_oracle_step_map(C.byref(self.p), C.byref(self.cs), C.byref(self.co),
                                          None if e is None else _ptr(e, _D), C.byref(self.cmap))
        else:
            rc = lib().ca_oracle_step(C.byref(self.p), C.byref(self.cs), C.byref(self.co),
                                      None if e is None else _ptr(e, _D))
        assert rc == 0
        return self.obs, self.rewards, self.game_over

    def set_map(self, static_map=None, rows=160, cols=160, cell=0.1, num_beams=512, num_to_store=3, max_range=6.0,
                range_res=0.1):
        """Map(16, 16, 0.1) of env.py:389-392 + a LaserScanSensor with its hard-coded parameters
        (LaserScanSensor.py:28-39)."""
        self.static_map = None if static_map is None else np.ascontiguousarray(static_map, np.uint8)
        self.cmap = OrcMap(None if self.static_map is None else _ptr(self.static_map, _U8), rows, cols, cell,
                           (rows * cell / 2.) / cell, (cols * cell / 2.) / cell)
        R = len(np.arange(0, max_range, range_res))
        self.scan_hist = np.full((self.E, self.N, num_to_store, num_beams), 255, np.uint8)
        self.scan = np.zeros((self.E, self.N, num_to_store, num_beams), np.float64)
        self.cscan = OrcScan(_ptr(self.scan_hist, _U8), _ptr(self.scan, _D), num_beams, num_to_store, R, -math.pi / 2,
                             math.pi / 2, range_res, max_range)

    def laserscan(self):
        rc = lib().ca_oracle_laserscan(C.byref(self.p), C.byref(self.cs), C.byref(self.cmap), C.byref(self.cscan))
        assert rc == 0
        return self.scan

    def rollout(self, table, n_steps, env_id_offset=0, case_stride=None):
        table = np.ascontiguousarray(table, np.float64)
        assert table.shape[1:] == (self.N, 6)
        stride = self.E if case_stride is None else case_stride
        rc = lib().ca_oracle_rollout(C.byref(self.p), C.byref(self.cs), C.byref(self.co), _ptr(table, _D),
                                     C.c_int32(table.shape[0]), C.c_int64(env_id_offset), C.c_int64(stride),
                                     C.c_int32(n_steps))
        assert rc == 0

    def rollout_ex(self, table, n_steps, env_id_offset=0, case_stride=None, ext_actions=None):
        """rollout() with external actions (GA3C-CADRL agents are queried here, once per step, like step()) and this
        oracle's static map: step + auto-reset for every workload bench.py times."""
        table = np.ascontiguousarray(table, np.float64)
        assert table.shape[1:] == (self.N, 6)
        stride = self.E if case_stride is None else case_stride
        for _ in range(int(n_steps)):
            e = ext_actions
            if (self.s["policy"] == POL_GA3C_CADRL).any():
                e = self.ga3c_query(e)
            e = None if e is None else np.ascontiguousarray(e, np.float64)
            rc = lib().ca_oracle_rollout_ex(C.byref(self.p), C.byref(self.cs), C.byref(self.co),
                                            None if e is None else _ptr(e, _D), _ptr(table, _D), C.c_int32(table.shape[0]),
                                            C.c_int64(env_id_offset), C.c_int64(stride), C.c_int32(1),
                                            None if self.cmap is None else C.byref(self.cmap))
            assert rc == 0

    def view(self, name):
        a = self.s[name]
        return a.reshape(self.E, self.N, *a.shape[1:])


def orca(pos, vel, pref, radius, max_speed, collab=0.5, time_horizon=5.0, time_step=0.1, max_neighbors=None,
         neighbor_dist=math.inf):
    """Batched rvo2 doStep velocities: float32 [E,N,2] x3, [E,N] x2 -> new_vel float32 [E,N,2]."""
    pos = np.ascontiguousarray(pos, np.float32)
    E, N = pos.shape[:2]
    vel, pref = np.ascontiguousarray(vel, np.float32), np.ascontiguousarray(pref, np.float32)
    radius, max_speed = np.ascontiguousarray(radius, np.float32), np.ascontiguousarray(max_speed, np.float32)
    out = np.zeros((E, N, 2), np.float32)
    rc = lib().ca_oracle_orca(C.c_int32(E), C.c_int32(N), _ptr(pos, _F), _ptr(vel, _F), _ptr(pref, _F),
                              _ptr(radius, _F), _ptr(max_speed, _F), C.c_float(collab), C.c_float(time_horizon),
                              C.c_float(time_step), C.c_int32(N if max_neighbors is None else max_neighbors),
                              C.c_float(neighbor_dist), _ptr(out, _F))
    assert rc == 0
    return out


ATAN2_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double)
SINCOS_FN = C.CFUNCTYPE(None, C.c_double, _D, _D)
_libm_keep = []


def set_libm(atan2_fn=None, sincos_fn=None):
    """Route the oracle's atan2 / (sin, cos) through Python callables (None = glibc): atan2_fn(y, x) -> float,
    sincos_fn(a) -> (sin, cos).  Process-wide; call set_libm() to restore."""
    a = ATAN2_FN(atan2_fn) if atan2_fn else None

    def _sc(x, ps, pc):
        sn, cs = sincos_fn(x)
        ps[0] = sn
        pc[0] = cs
    b = SINCOS_FN(_sc) if sincos_fn else None
    _libm_keep[:] = [a, b]
    L = lib()
    L.ca_oracle_set_libm.restype = None
    L.ca_oracle_set_libm(C.cast(a, C.c_void_p) if a else None, C.cast(b, C.c_void_p) if b else None)


def set_tie_order(reverse=False):
    """tests only: exactly tied distSq neighbours in reverse visit order (orca_ref.h g_tie_reverse); process-wide"""
    L = lib()
    L.ca_oracle_set_tie_order.restype = None
    L.ca_oracle_set_tie_order(C.c_int(1 if reverse else 0))


def round2(x):
    return lib().ca_oracle_round2(float(x))
