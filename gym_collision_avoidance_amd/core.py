"""BatchedSim: the (num_envs x num_agents) simulator state as PyTorch-ROCm tensors + the calls into libcagpu.so.

torch is plumbing here (device memory, streams); every per-step computation happens in the HIP kernels of
csrc/cagpu.hip through the C ABI of include/cagpu.h.  Layout: agent-major SoA, index e*N + a.
Reference objects this replaces: the list[Agent] owned by CollisionAvoidanceEnv (collision_avoidance_env.py:141,
agent.py:29-138) and the per-step loops of collision_avoidance_env.py:156-234.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _native as nat

_F64 = ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "goal_x", "goal_y", "radius", "pref_speed",
        "time_remaining", "t", "slt", "ep_reward", "turning_dir")
STAT_NAMES = ("episodes", "collision_episodes", "all_at_goal_episodes", "stuck_episodes", "sum_steps",
              "sum_total_reward", "sum_time_to_goal", "sum_extra_time_to_goal")


def make_params(num_envs, num_agents, max_obs=None, dt=0.1, max_time_ratio=8.0, sort_mode=nat.SORT_CLOSEST_FIRST,
                game_over_mode=nat.OVER_ALL_DONE, rvo_max_neighbors=None, near_goal_threshold=0.2,
                getting_close_range=0.2, sensing_horizon=math.inf, reward_at_goal=1.0, reward_collision=-0.25,
                reward_time_step=0.0, reward_wiggly=0.0, wiggly_threshold=math.inf, reward_min=None, reward_max=None,
                rvo_time_horizon=5.0, rvo_collab_coeff=0.5, max_heading_change=math.pi / 3, obs_clip=None,
                reward_collision_wall=-0.25, rvo_dt=None, ragged=0):
    """CaParams with the reference's Config defaults (config.py:28-86) for an EvaluateConfig-style run."""
    p = nat.CaParams()
    p.num_envs, p.num_agents = int(num_envs), int(num_agents)
    p.max_obs = int(num_agents - 1 if max_obs is None else max_obs)
    p.obs_clip = p.max_obs if obs_clip is None else int(obs_clip)
    p.ragged = int(ragged)   # envs may hold fewer than num_agents agents (case rows with radius <= 0 = empty slots)
    p.sort_mode, p.game_over_mode = int(sort_mode), int(game_over_mode)
    p.rvo_max_neighbors = int(num_agents if rvo_max_neighbors is None else rvo_max_neighbors)
    p.dt, p.near_goal_threshold, p.max_time_ratio = dt, near_goal_threshold, max_time_ratio
    p.getting_close_range, p.sensing_horizon = getting_close_range, sensing_horizon
    p.reward_at_goal, p.reward_collision, p.reward_time_step = reward_at_goal, reward_collision, reward_time_step
    p.reward_wiggly, p.wiggly_threshold = reward_wiggly, wiggly_threshold
    # collision_avoidance_env.py:589-599: clip bounds = min/max of the possible reward values
    vals = [reward_at_goal, reward_collision, reward_time_step, reward_collision_wall, reward_wiggly]
    p.reward_min = min(vals) if reward_min is None else reward_min
    p.reward_max = max(vals) if reward_max is None else reward_max
    p.rvo_time_horizon, p.rvo_collab_coeff = rvo_time_horizon, rvo_collab_coeff
    p.max_heading_change = max_heading_change
    p.reward_collision_wall = reward_collision_wall
    p.rvo_dt = dt if rvo_dt is None else rvo_dt   # RVOPolicy.py:13: Config.DT, whatever dt a later step() is called with
    return p


GA3C_DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ga3c_cadrl", "IROS18",
                                    "network_01900000.npz")


# the current stream's raw handle without building a torch.cuda.Stream object around it (1.2 -> 0.3 us on the way to a launch)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class BatchedSim(object):
    PROBE_EVERY = 8   # the device's fault word is probed behind every 8th ring refill (and every 256th single launch)

    def __init__(self, params, device="cuda:0", record_actions=False, pipeline=True):
        """pipeline: hand the kernels CaState.next_action (include/cagpu.h): the step kernel then computes the RVO policy of
        the NEXT step beside the sensing half of this one and the next launch starts at the move -- bit-identical results.
        Code that writes `state[...]` tensors directly must call invalidate_plan() afterwards (or overwrite `flags`
        with words whose PLAN_VALID bit is clear); reset() does it by itself."""
        if not torch.cuda.is_available():
            raise nat.CagpuError("BatchedSim needs a ROCm device: the hot path is HIP-only (no CPU fallback)")
        self.lib = nat.lib()
        self.p = params
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else 0
        E, N, K = params.num_envs, params.num_agents, params.max_obs
        self.E, self.N, self.K, self.W = E, N, K, 6 + 7 * K
        dev = self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        # The whole simulator state lives in ONE slab (every array a 256-byte-aligned view of it): the look-ahead ring
        # (step_lookahead) snapshots it with one device copy before it runs ahead and restores it the same way when the
        # caller turns out to need the state of a step already handed out.
        specs = [(n, (E, N), torch.float64) for n in _F64] + [
            ("last_action", (E, N, 2), torch.float32), ("flags", (E, N), torch.int32),   # (flags: uint32 bit pattern)
            ("step_num", (E, N), torch.int32), ("episode_step", (E,), torch.int32), ("reset_count", (E,), torch.int32),
            ("env_stats", (E, 8), torch.float64), ("next_action", (E, N, 4), torch.float32)]
        offs, total = [], 0
        for _, shape, dt in specs:
            offs.append(total)
            total += (int(np.prod(shape)) * torch.empty((), dtype=dt).element_size() + 255) // 256 * 256
        self._slab = torch.zeros((total,), dtype=torch.uint8, device=dev)
        self._state = {}
        for (n, shape, dt), off in zip(specs, offs):
            nb = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self._state[n] = self._slab[off:off + nb].view(dt).view(shape)
        self._la = None           # the look-ahead ring (enable_lookahead)
        self.pipeline = bool(pipeline)
        self._obs = z((E, N, self.W), torch.float32)
        self._rewards = z((E, N), torch.float32)
        self._done = z((E, N), torch.uint8)
        self._game_over = z((E,), torch.uint8)
        self.actions = z((E, N, 2), torch.float32) if record_actions else None
        self.orca_vel = z((E, N, 2), torch.float32) if record_actions else None
        self._cs = nat.CaState(**{n: self._state[n].data_ptr() for n in nat.STATE_FIELDS if n in self._state})
        self._rvo = None          # RVOPolicy's stochastic branches (set_rvo_stochastic)
        self._variants = []       # per-agent sensor arguments beyond the primary pair (set_sensor_variants)
        if not self.pipeline:
            self._cs.next_action = None
        self._p_ref, self._cs_ref = C.byref(self.p), C.byref(self._cs)   # (cached: the structs live as long as the sim)
        self._co = nat.CaOut(obs=self._obs.data_ptr(), rewards=self._rewards.data_ptr(), done=self._done.data_ptr(),
                             game_over=self._game_over.data_ptr(),
                             actions=self.actions.data_ptr() if record_actions else None,
                             orca_vel=self.orca_vel.data_ptr() if record_actions else None)
        # more than 64 agents per env: the large-env kernel's per-pair columns live in a workspace (include/cagpu.h)
        self._workspace = None
        wsb = int(self.lib.cagpu_workspace_bytes(C.byref(self.p)))
        if wsb:
            self._workspace = torch.empty((wsb,), dtype=torch.uint8, device=dev)
            self._co.workspace, self._co.workspace_bytes = self._workspace.data_ptr(), wsb
        # fresh_outputs: every step / rollout writes obs, rewards, done and game_over into NEWLY allocated tensors (the
        # previous ones stay valid and belong to whoever holds them: the env API's "fresh arrays every step" without a
        # copy kernel).  Every element of the four outputs is rewritten by every step launch
        # (tests/test_gpu_parity.py::test_step_rewrites_every_output_element), so nothing is carried over in them.
        self.fresh_outputs = False
        self._ar = None
        self._fast_args = None    # prebuilt ctypes arguments of the external-action-free step (see step())
        self._table = None
        self._keep = []
        self._map = None
        self._scan = None
        self.scan = None
        self.ga3c_fused = False   # cagpu_ga3c with obs = NULL: sensing fused into the network kernel (see ga3c())
        self._net = None          # GA3C-CADRL weights (load_ga3c); _nets: {checkpoint index: (CaNet, tensors)}
        self._net_tensors = None
        self._nets = {}
        self._agent_net = None    # int32 [E, N]: which checkpoint an agent runs (set_ga3c_assignment)
        self._has_ga3c = False    # some agent's policy is CA_POL_GA3C_CADRL (set_plugins)
        self._ga3c_ext = None
        self.ga3c_logits = None
        self._fault = None        # the non-blocking fault-word probe of the product path (_fault_probe)
        # (the first probe -- it creates the side stream and the pinned word: milliseconds -- on the second launch of a
        # simulator's life, not 256 launches in, in the middle of somebody's timed loop)
        self._steps_since_probe = 254

    # ---------------------------------------------------------------- what the outside reads
    # `state` and the four outputs are those of the step last handed out: reading them goes through sync(), which rewinds a
    # look-ahead ring that has run ahead (a no-op without one); the methods of this class use the underscored names.
    @property
    def state(self):
        self.sync()
        return self._state

    def _synced(name):   # noqa: N805 -- builds the four output properties
        def get(self):
            self.sync()
            return getattr(self, name)

        def put(self, value):
            setattr(self, name, value)
        return property(get, put)
    obs, rewards, done, game_over = _synced("_obs"), _synced("_rewards"), _synced("_done"), _synced("_game_over")
    del _synced

    # ---------------------------------------------------------------- plumbing
    def _stream(self):
        return C.c_void_p(_raw_stream(self._dev_index) if _raw_stream else torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype):
        if x is None:
            return None
        t = torch.as_tensor(x, dtype=dtype)
        if t.device != self.device:
            t = t.to(self.device)
        return t.contiguous()

    def update_params(self, **changes):
        """Change CaParams fields (e.g. rvo_collab_coeff=0.3) on a live sim.  A pipelined plan -- CaState.next_action of
        every env and the fixture table's reset_plan / reset_obs -- was computed under the parameters in force when it
        was made (RVO horizon / collaboration / dt, sensing horizon, neighbour count, the observation layout), so it is
        forgotten here and the table's reset rows are recomputed.  Writing `sim.p.<field>` directly skips this: call
        update_params() (or invalidate_plan() + set_fixture_table()) instead."""
        self.sync()
        for k_, v in changes.items():
            if k_ in ("num_envs", "num_agents", "max_obs"):
                raise ValueError("%s is fixed at construction (tensor shapes)" % k_)
            if not hasattr(self.p, k_):
                raise AttributeError("CaParams has no field %r" % k_)
            setattr(self.p, k_, v)
        self._fast_args = None
        if self._la is not None:
            self._la["in_kernel"].clear()
            self._la["prep"] = None
        self.invalidate_plan()
        if self._table is not None:
            ar = self._ar
            self.set_fixture_table(self._table, env_id_offset=ar.env_id_offset, case_stride=ar.case_stride,
                                   heading_seed=ar.heading_seed)

    def set_rvo_stochastic(self, heading_noise=None, collab_coeff=None, anti_collab_t=1.0, seed=0, noise_std=0.5):
        """RVOPolicy's two stochastic branches for a whole batch, drawn on the DEVICE (torch's Philox generator) right
        before every step launch and handed to the kernel as CaState.rvo_collab / rvo_heading_noise:
          * heading_noise: bool mask broadcastable to [E, N] -- agents whose policy has `heading_noise` set get
            N(0, noise_std) added to their delta heading each query (policies/RVOPolicy.py:118-119);
          * collab_coeff < 0: anti-collaborative agents (RVOPolicy.py:77-88) -- every agent carries the policy object's
            `use_non_coop_policy` (initially True); whenever its clock is within DT of a multiple of anti_collab_t
            (`round(t % T, 3) < DT or round(T - t % T, 3) < DT`) it is redrawn, True with probability 1 - |c|; the ego's
            collaboration coefficient of the query is 0 while it is True, c otherwise.
        Both None: switched off (the deterministic kernels, pipelined plan included).  The reference draws from numpy's
        global stream, one agent after the other: the batched draws are the same distributions, not the same numbers."""
        self.sync()
        self._fast_args = None
        # (whichever branch is not configured below must not keep pointing at a tensor of an earlier configuration)
        self._cs.rvo_collab = None
        self._cs.rvo_heading_noise = None
        if heading_noise is None and (collab_coeff is None or collab_coeff >= 0):
            self._rvo = None
            self._cs.rvo_collab = None
            self._cs.rvo_heading_noise = None
            return
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        mask = None
        if heading_noise is not None:
            mask = torch.from_numpy(np.array(np.broadcast_to(np.asarray(heading_noise, bool), (self.E, self.N)))).to(self.device)
        self._rvo = dict(gen=gen, mask=mask, std=float(noise_std), T=float(anti_collab_t),
                         c=None if (collab_coeff is None or collab_coeff >= 0) else float(collab_coeff),
                         non_coop=torch.ones((self.E, self.N), dtype=torch.bool, device=self.device),  # RVOPolicy.py:33
                         collab=None, noise=None)
        self.invalidate_plan()

    def _rvo_draw(self):
        """this step's draws (see set_rvo_stochastic): a handful of small device ops, no host synchronisation"""
        r = self._rvo
        if r["c"] is not None:
            t, T, dt = self._state["t"], r["T"], float(self.p.rvo_dt)
            tm = torch.remainder(t, T)
            r3 = lambda x: torch.round(x * 1000.0) / 1000.0        # numpy's scalar round(x, 3)
            redraw = (r3(tm) < dt) | (r3(T - tm) < dt)
            u = torch.rand((self.E, self.N), generator=r["gen"], device=self.device, dtype=torch.float64)
            r["non_coop"] = torch.where(redraw, u < (1.0 - abs(r["c"])), r["non_coop"])   # np.random.choice([True, False], p=[1 - |c|, |c|])
            r["collab"] = torch.where(r["non_coop"], 0.0, r["c"]).to(torch.float32).contiguous()
            self._cs.rvo_collab = r["collab"].data_ptr()
        if r["mask"] is not None:
            z = torch.randn((self.E, self.N), generator=r["gen"], device=self.device, dtype=torch.float64)
            r["noise"] = (z * r["std"] * r["mask"]).contiguous()
            self._cs.rvo_heading_noise = r["noise"].data_ptr()

    def set_sensor_variants(self, variants=None):
        """Agents of one batch whose OtherAgentsStatesSensor arguments differ (the reference gives every agent its own
        sensor object: `sensor.set_args({'agent_sorting_method': ..., 'max_num_other_agents_observed': ...})`,
        sensors/Sensor.py:19-23).  The kernels take ONE obs_clip / sort_mode per launch -- CaParams holds the primary pair --
        so every further pair costs one cagpu_observe launch per step: `variants` = [(mask, obs_clip, sort_mode), ...] with
        mask a bool array broadcastable to [E, N]; after every step / reset / observe the rows of the masked agents are
        replaced by their rows under that pair.  None / []: switched off."""
        self.sync()
        self._variants = []
        for mask, clip, sort in (variants or []):
            m = torch.from_numpy(np.array(np.broadcast_to(np.asarray(mask, bool), (self.E, self.N)))).to(self.device)
            if not (0 <= int(clip) <= self.K):
                raise ValueError("obs_clip %d outside [0, %d]" % (clip, self.K))
            self._variants.append((m.unsqueeze(-1), int(clip), int(sort), torch.empty_like(self._obs)))

    def _apply_sensor_variants(self):
        if not self._variants:
            return
        p, co = self.p, self._co
        keep = (p.obs_clip, p.sort_mode, co.obs)
        try:
            for m, clip, sort, buf in self._variants:
                p.obs_clip, p.sort_mode, co.obs = clip, sort, buf.data_ptr()
                nat.check(self.lib.cagpu_observe(C.byref(p), C.byref(self._cs), C.byref(co), self._stream()))
                torch.where(m, buf, self._obs, out=self._obs)   # (elementwise, no host synchronisation)
        finally:
            p.obs_clip, p.sort_mode, co.obs = keep

    def invalidate_plan(self):
        """Forget the pipelined policy query (CaState.next_action): call after writing state tensors directly."""
        self.sync()
        self._state["flags"].bitwise_and_(~nat.PLAN_VALID)

    # ---------------------------------------------------------------- configuration
    def set_plugins(self, policy, dynamics=None, is_learning=None, still_learning=None):
        """policy / dynamics: int ids (CA_POL_*, CA_DYN_*), broadcastable to [E,N].  The learning bits default to
        what the reference's policy classes set (LearningPolicy.py:9-11: str == 'learning', is_still_learning)."""
        self.sync()
        E, N = self.E, self.N
        pol = np.broadcast_to(np.asarray(policy, np.int64), (E, N))
        dyn = np.broadcast_to(np.asarray(0 if dynamics is None else dynamics, np.int64), (E, N))
        learn = (pol == nat.POL_LEARNING) | (pol == nat.POL_LEARNING_GA3C)
        isl = learn if is_learning is None else np.broadcast_to(np.asarray(is_learning, bool), (E, N))
        stl = learn if still_learning is None else np.broadcast_to(np.asarray(still_learning, bool), (E, N))
        bits = (pol << nat.POLICY_SHIFT) | (dyn << nat.DYNAMICS_SHIFT) | (isl * nat.IS_LEARNING) | \
               (stl * nat.STILL_LEARNING)
        cur = self._state["flags"]
        cur.copy_((cur & (0x3F | nat.ABSENT)) | torch.as_tensor(bits.astype(np.int32), device=self.device))
        self._has_ga3c = bool((pol == nat.POL_GA3C_CADRL).any())

    def ga3c_rows(self):
        """number of agents the last ga3c() call evaluated (device -> host read: synchronises)"""
        return int(self._net_tensors["rows_scratch"][self.E * self.N].item())

    def load_ga3c(self, weights=None, keep_logits=False, index=0):
        """Upload the GA3C-CADRL network (GA3CCADRLPolicy.initialize_network, GA3CCADRLPolicy.py:23-47).  `weights`:
        an .npz written by oracle/extract_ga3c_weights.py (default: the shipped IROS18/network_01900000, the
        reference's default checkpoint) or a dict of float32 arrays with the same keys.
        `index`: several checkpoints may be loaded side by side (index 0, 1, ...); set_ga3c_assignment() says which agent
        runs which (the reference gives every agent its own policy object and session).  Without an assignment every
        GA3C-CADRL agent runs checkpoint 0."""
        self.sync()
        if weights is None:
            weights = GA3C_DEFAULT_WEIGHTS
        if isinstance(weights, str):
            with np.load(weights) as z:
                weights = {k: z[k] for k in z.files}
        names = {"logits_kernel": "logits_p_kernel", "logits_bias": "logits_p_bias"}
        shapes = {"lstm_kernel": (71, 256), "lstm_bias": (256,), "layer1_kernel": (68, 256), "layer1_bias": (256,),
                  "layer2_kernel": (256, 256), "layer2_bias": (256,), "fc1_kernel": (256, 256), "fc1_bias": (256,),
                  "logits_kernel": (256, 11), "logits_bias": (11,), "input_mean": (138,), "input_std": (138,)}
        ts = {}
        for f in nat.NET_FIELDS:
            a = np.ascontiguousarray(weights[names.get(f, f)], dtype=np.float32)
            if a.shape != shapes[f]:
                raise ValueError("GA3C-CADRL weight %s has shape %s, expected %s" % (f, a.shape, shapes[f]))
            ts[f] = torch.from_numpy(a).to(self.device)
        # scratch of cagpu_ga3c: the packed list of the agents that need an action this step (+ their count)
        ts["rows_scratch"] = torch.empty((self.E * self.N + 6,), dtype=torch.int32, device=self.device)
        # the four big matrices as fp16 planes in matrix-core fragment order: split once per checkpoint on the device
        ts["packed"] = torch.empty((int(self.lib.cagpu_ga3c_packed_bytes()),), dtype=torch.uint8, device=self.device)
        net = nat.CaNet(**{f: ts[f].data_ptr() for f in nat.NET_FIELDS + ("rows_scratch", "packed")})
        nat.check(self.lib.cagpu_ga3c_pack(C.byref(net), ts["packed"].data_ptr(), ts["packed"].numel(), self._stream()))
        self._nets[int(index)] = (net, ts)
        if int(index) == 0 or self._net is None:
            self._net_tensors, self._net = ts, net
        if keep_logits or self.ga3c_logits is None:
            self.ga3c_logits = torch.zeros((self.E, self.N, 11), dtype=torch.float32, device=self.device) \
                if keep_logits else None

    def set_ga3c_assignment(self, index):
        """index: int array broadcastable to [E, N] -- the checkpoint (load_ga3c(..., index=)) each GA3C-CADRL agent runs;
        None: everybody runs checkpoint 0."""
        if index is None:
            self._agent_net = None
            return
        a = np.array(np.broadcast_to(np.asarray(index, np.int32), (self.E, self.N)))   # (a writable copy for torch.from_numpy)
        missing = set(np.unique(a).tolist()) - set(self._nets)
        if missing:
            raise nat.CagpuError("GA3C-CADRL checkpoint index %s assigned but not loaded" % sorted(missing))
        self._agent_net = torch.from_numpy(a).to(self.device)

    def ga3c(self, ext=None, fused=None):
        """Query the network for every live GA3C-CADRL agent on the CURRENT observation; the action indices land in
        `ext[..., 0]` (a float64 [E,N,2] tensor, allocated here if not given), which step() then consumes.
        fused (default: self.ga3c_fused): the kernel computes the ego-centric observation of the agents it evaluates from
        the state arrays itself (cagpu_ga3c with obs = NULL: "obs + network inference fused in-kernel") instead of reading
        the rows the last step stored -- same bits; needs num_agents <= 32, no time_to_impact sorting and no per-agent
        sensor variants."""
        self.sync()
        if self._net is None:
            raise nat.CagpuError("GA3C-CADRL agents present but no network loaded: call load_ga3c() "
                                 "(policy.initialize_network() in the env API)")
        if ext is None:
            if self._ga3c_ext is None:
                self._ga3c_ext = torch.zeros((self.E, self.N, 2), dtype=torch.float64, device=self.device)
            ext = self._ga3c_ext
        lg = None if self.ga3c_logits is None else self.ga3c_logits.data_ptr()
        fused = self.ga3c_fused if fused is None else fused
        if fused and (self.N > 32 or self.p.sort_mode == nat.SORT_TIME_TO_IMPACT or self._variants):
            fused = False
        obs_ptr = None if fused else self._obs.data_ptr()
        if self._agent_net is None:      # one checkpoint (index 0) for every GA3C-CADRL agent
            nat.check(self.lib.cagpu_ga3c(C.byref(self.p), C.byref(self._cs), obs_ptr, C.byref(self._net),
                                          ext.data_ptr(), lg, self._stream()))
            return ext
        for idx in sorted(self._nets):   # one launch per checkpoint, each over its own agents (CaNet.agent_net / net_index)
            net = self._nets[idx][0]
            net.agent_net, net.net_index = self._agent_net.data_ptr(), idx
            nat.check(self.lib.cagpu_ga3c(C.byref(self.p), C.byref(self._cs), obs_ptr, C.byref(net),
                                          ext.data_ptr(), lg, self._stream()))
            net.agent_net = None
        return ext

    def generate_cases(self, num_cases, seed, side_length=4.0, speed_bnds=(0.5, 2.0), radius_bnds=(0.2, 0.8),
                       return_status=False, num_agents=None, return_counts=False):
        """`num_cases` random scenarios for this sim's agent count, drawn ON THE DEVICE by cagpu_generate_cases
        (generate_rand_test_case_multi behind test_cases.get_testcase_random, test_cases.py:212-253): float64 device
        tensor [num_cases, N, 6], ready for reset() / set_fixture_table().  `side_length`: a number, or (lo, hi) to draw
        it per case.  Same (seed, case index) -> same scenario.

        `num_agents=(lo, hi)`: a RAGGED table (cagpu_generate_cases_ragged) -- the agent count of every case drawn in
        lo .. hi <= N like the reference's num_agents=None (test_cases.py:224-227), rows past it zero (empty slots; the
        sim must be built with ragged=1); `side_length` is then a number, (lo, hi), or the reference's list of
        {"num_agents": [lo, hi), "side_length": [lo, hi]} dicts (config.py:118-131)."""
        out = torch.empty((int(num_cases), self.N, 6), dtype=torch.float64, device=self.device)
        status = torch.zeros((int(num_cases),), dtype=torch.int32, device=self.device)
        counts = None
        sp = (float(speed_bnds[0]), float(speed_bnds[1]), float(radius_bnds[0]), float(radius_bnds[1]))
        if num_agents is None and not isinstance(side_length, list):
            lo, hi = (side_length, side_length) if np.isscalar(side_length) else side_length
            nat.check(self.lib.cagpu_generate_cases(int(num_cases), self.N, float(lo), float(hi), *sp,
                                                    int(seed) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(), status.data_ptr(),
                                                    self._stream()))
        else:
            n_lo, n_hi = (self.N, self.N) if num_agents is None else (int(num_agents[0]), int(num_agents[1]))
            if isinstance(side_length, list):
                rg = [[c["num_agents"][0], c["num_agents"][1], c["side_length"][0], c["side_length"][1]]
                      for c in side_length]
            else:
                lo, hi = (side_length, side_length) if np.isscalar(side_length) else side_length
                rg = [[0, 1 << 30, lo, hi]]
            rg = np.ascontiguousarray(rg, dtype=np.float64)
            counts = torch.zeros((int(num_cases),), dtype=torch.int32, device=self.device)
            nat.check(self.lib.cagpu_generate_cases_ragged(int(num_cases), self.N, n_lo, n_hi, rg.ctypes.data, len(rg), *sp,
                                                           int(seed) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(),
                                                           counts.data_ptr(), status.data_ptr(), self._stream()))
        res = (out,) + ((status,) if return_status else ()) + ((counts,) if return_counts else ())
        return res if len(res) > 1 else out

    def set_fixture_table(self, table, env_id_offset=0, case_stride=None, heading_seed=0):
        """Enable DummyVecEnv-style auto-reset from a fixture table [C,N,6] (vec_env.py:120-128,
        test_cases.py:593-624): env e's k-th reset loads case (env_id_offset + e + k*case_stride) % C.
        heading_seed != 0: training mode (test_cases.py:558-559) -- an auto-reset draws the initial heading uniformly in
        [-pi, pi) on the device (Philox of seed, global env id, reset count, agent) instead of pointing at the goal."""
        self.sync()
        self._fast_args = None
        if self._la is not None:
            self._la["in_kernel"].clear()
            self._la["prep"] = None
        if table is None:
            self._ar, self._table = None, None
            return
        t = self._dev(table, torch.float64)
        assert t.dim() == 3 and t.shape[1:] == (self.N, 6), t.shape
        self._table = t
        # the reset observation of every case, computed once by the reset kernel itself on a scratch batch of C envs
        C_ = int(t.shape[0])
        ps = nat.CaParams.from_buffer_copy(self.p)
        ps.num_envs = C_
        scratch = BatchedSim(ps, device=self.device, pipeline=self.pipeline)
        scratch.reset(t)
        self._reset_obs = scratch.obs
        # ... and, for the pipelined step kernel, the first action of every case (CaAutoReset.reset_plan): an RVO agent's
        # plan depends on positions / velocities / radii only, so the scratch batch may take every slot for an RVO agent
        self._reset_plan = None
        if self.pipeline and scratch.try_plan():
            self._reset_plan = scratch.state["next_action"]
        torch.cuda.current_stream(self.device).synchronize()
        self._ar = nat.CaAutoReset(table=t.data_ptr(), n_cases=C_, env_id_offset=int(env_id_offset),
                                   case_stride=int(self.E if case_stride is None else case_stride),
                                   reset_obs=self._reset_obs.data_ptr(),
                                   reset_plan=None if self._reset_plan is None else self._reset_plan.data_ptr(),
                                   heading_seed=int(heading_seed) & 0xFFFFFFFFFFFFFFFF)

    # ---------------------------------------------------------------- the C-ABI calls
    def reset(self, cases, headings=None, mask=None):
        self.sync()   # (a ring that ran ahead is rewound even for a reset of EVERY env: env_stats outlive a reset and must not
        #                hold episodes of steps that were never handed out)
        if self.fresh_outputs:   # the tensors the last step() handed out belong to their holder: never written again
            self._new_outputs(keep=mask is not None)
        c = self._dev(cases, torch.float64)
        assert tuple(c.shape) == (self.E, self.N, 6), c.shape
        h = self._dev(headings, torch.float64)
        m = self._dev(mask, torch.uint8)
        nat.check(self.lib.cagpu_reset(C.byref(self.p), C.byref(self._cs), C.byref(self._co), c.data_ptr(),
                                       None if h is None else h.data_ptr(), None if m is None else m.data_ptr(),
                                       self._stream()))
        self._keep = [c, h, m]  # keep alive until the stream has consumed them
        self._apply_sensor_variants()
        return self._obs

    def reset_from_table(self, env_id_offset=None):
        """Initial load: env e <- case (env_id_offset + e) % C of the fixture table."""
        assert self._table is not None
        off = self._ar.env_id_offset if env_id_offset is None else env_id_offset
        idx = (torch.arange(self.E, device=self.device) + off) % self._table.shape[0]
        return self.reset(self._table[idx])

    def set_map(self, static_map=None, rows=160, cols=160, cell=0.1, num_beams=512, num_to_store=3, max_range=6.0,
                range_res=0.1, min_angle=-math.pi / 2, max_angle=math.pi / 2):
        """Static occupancy grid (Map.py:6-24; bool [rows, cols], True = occupied, or None for an empty map) + the
        LaserScanSensor buffers with the reference's hard-coded parameters (LaserScanSensor.py:28-39).  With a map
        set, step() also tests wall collisions (collision_avoidance_env.py:494-506)."""
        self.sync()
        bits = None
        self._fast_args = None
        if static_map is not None:
            m = np.asarray(static_map).astype(bool)
            assert m.shape == (rows, cols), m.shape
            wpr = (cols + 31) // 32
            pad = np.zeros((rows, wpr * 32), dtype=np.uint8)
            pad[:, :cols] = m
            words = np.packbits(pad.reshape(rows, wpr, 32), axis=-1, bitorder="little").view(np.uint32).reshape(rows, wpr)
            bits = torch.from_numpy(words.view(np.int32).copy()).to(self.device)
        self._map_bits = bits
        self._map = nat.CaMap(static_bits=None if bits is None else bits.data_ptr(), rows=rows, cols=cols, cell=cell,
                              origin_r=(rows * cell / 2.) / cell, origin_c=(cols * cell / 2.) / cell)
        R = len(np.arange(0, max_range, range_res))
        self.scan_hist = torch.full((self.E, self.N, num_to_store, num_beams), 255, dtype=torch.uint8,
                                    device=self.device)
        self.scan = torch.zeros((self.E, self.N, num_to_store, num_beams), dtype=torch.float32, device=self.device)
        self._scan = nat.CaScan(hist=self.scan_hist.data_ptr(), out=self.scan.data_ptr(), num_beams=num_beams,
                                num_to_store=num_to_store, num_ranges=R, min_angle=min_angle, max_angle=max_angle,
                                range_res=range_res, max_range=max_range)

    def laserscan(self):
        """'laserscan' observation [E,N,num_to_store,num_beams] of the current state (call after reset / step)."""
        assert self._map is not None, "set_map() first"
        self.sync()
        nat.check(self.lib.cagpu_laserscan(C.byref(self.p), C.byref(self._cs), C.byref(self._map),
                                           C.byref(self._scan), self._stream()))
        return self.scan

    def _new_outputs(self, keep=False):
        """keep: the new tensors start as copies of the current ones (a masked reset rewrites only some envs' rows)"""
        co = self._co
        new = (lambda t: t.clone()) if keep else torch.empty_like
        self._obs = new(self._obs); co.obs = self._obs.data_ptr()
        self._rewards = new(self._rewards); co.rewards = self._rewards.data_ptr()
        self._done = new(self._done); co.done = self._done.data_ptr()
        self._game_over = new(self._game_over); co.game_over = self._game_over.data_ptr()

    def step(self, ext_actions=None, ext_state=None):
        """ext_state: float64 [E, N, 5] = px, py, vx, vy, heading for agents with ExternalDynamics whose motion of THIS step
        was integrated outside (a user Dynamics subclass on the host; NaN rows: none) -- applied by the kernel at the move
        (CaState.ext_state); None: nobody."""
        if self._la is not None:
            self.sync()
        if self._rvo is not None:
            self._rvo_draw()
        if ext_state is not None or self._cs.ext_state:
            self._ext_state = self._dev(ext_state, torch.float64)
            if self._ext_state is not None:
                assert tuple(self._ext_state.shape) == (self.E, self.N, 5), self._ext_state.shape
            self._cs.ext_state = None if self._ext_state is None else self._ext_state.data_ptr()
        if ext_actions is None and not self._has_ga3c:
            # env.step(None) with built-in policies only (env_utils.py:50): the per-step host path is one ctypes call
            # with prebuilt arguments -- at ~20 us per launch the interpreter is otherwise on the critical path
            fa = self._fast_args
            if fa is None:
                ar = None if self._ar is None else C.byref(self._ar)
                if self._map is not None:
                    fa = (self.lib.cagpu_step_map, (C.byref(self.p), C.byref(self._cs), C.byref(self._co), None, ar,
                                                    C.byref(self._map)))
                else:
                    fa = (self.lib.cagpu_step, (C.byref(self.p), C.byref(self._cs), C.byref(self._co), None, ar))
                self._fast_args = fa
            if self.fresh_outputs:
                self._new_outputs()
            rc = fa[0](*fa[1], _raw_stream(self._dev_index) if _raw_stream else torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                nat.check(rc)
            self._steps_since_probe += 1
            if self._steps_since_probe >= 256:
                self._steps_since_probe = 0
                self._fault_probe()
            if self._variants:
                self._apply_sensor_variants()
            return self._obs, self._rewards, self._game_over
        e = self._dev(ext_actions, torch.float64)
        if e is not None:
            assert tuple(e.shape) == (self.E, self.N, 2), e.shape
        if self._has_ga3c:  # policy query on the pre-step observation (collision_avoidance_env.py:319-323)
            if e is not None:  # the caller's external actions travel in the same buffer; never write into theirs
                if self._ga3c_ext is None:
                    self._ga3c_ext = torch.zeros((self.E, self.N, 2), dtype=torch.float64, device=self.device)
                self._ga3c_ext.copy_(e)
            e = self.ga3c(None if e is None else self._ga3c_ext)
        if self.fresh_outputs:
            self._new_outputs()
        if self._map is not None:
            nat.check(self.lib.cagpu_step_map(C.byref(self.p), C.byref(self._cs), C.byref(self._co),
                                              None if e is None else e.data_ptr(),
                                              None if self._ar is None else C.byref(self._ar), C.byref(self._map),
                                              self._stream()))
        else:
            nat.check(self.lib.cagpu_step(C.byref(self.p), C.byref(self._cs), C.byref(self._co),
                                          None if e is None else e.data_ptr(),
                                          None if self._ar is None else C.byref(self._ar), self._stream()))
        self._keep = [e]
        self._apply_sensor_variants()
        return self._obs, self._rewards, self._game_over

    def rollout(self, n_steps, ext_actions=None):
        self.sync()
        # (CaState.ext_state belongs to the ONE step() call that was given it: never re-applied by the steps of a rollout)
        self._cs.ext_state, self._ext_state = None, None
        if self._has_ga3c or self._rvo is not None:  # the network / the stochastic RVO draws run between steps
            for _ in range(int(n_steps)):
                self.step(ext_actions)
            return self._obs, self._rewards, self._game_over
        e = self._dev(ext_actions, torch.float64)
        if self.fresh_outputs:
            self._new_outputs()
        nat.check(self.lib.cagpu_rollout(C.byref(self.p), C.byref(self._cs), C.byref(self._co),
                                         None if e is None else e.data_ptr(),
                                         None if self._ar is None else C.byref(self._ar), int(n_steps),
                                         self._stream()))
        self._keep = [e]
        self._apply_sensor_variants()
        return self._obs, self._rewards, self._game_over

    def try_plan(self):
        """cagpu_plan: the policy query of the next step ahead of time (fills state['next_action'], sets PLAN_VALID).
        Returns False where the pipelined kernel has no instantiation for this batch (the step kernels then query the
        policy at the start of the step, as without next_action)."""
        self.sync()
        rc = self.lib.cagpu_plan(C.byref(self.p), C.byref(self._cs), self._stream())
        if rc == nat.CA_EUNSUPPORTED:
            return False
        nat.check(rc)
        return True

    def observe(self):
        self.sync()
        if self.fresh_outputs:   # (cagpu_observe rewrites obs only: the other outputs carry over)
            self._new_outputs(keep=True)
        nat.check(self.lib.cagpu_observe(C.byref(self.p), C.byref(self._cs), C.byref(self._co), self._stream()))
        self._apply_sensor_variants()
        return self._obs

    # ---------------------------------------------------------------- the look-ahead ring behind env.step(None)
    def lookahead_ok(self):
        """can step_lookahead() run this batch?  Every policy has to be answered inside the step kernel and nothing may
        happen BETWEEN two steps on the host or in another kernel: no GA3C-CADRL network, no per-step stochastic RVO
        draws, no per-agent sensor variants (extra cagpu_observe launches), no static map (wall test + laser scan)."""
        return not (self._has_ga3c or self._rvo is not None or self._variants or self._map is not None)

    def enable_lookahead(self, k, fresh=True, adaptive=False, start=None):
        """Serve step(None) from a ring of `k` steps computed ahead of time in ONE launch (cagpu_rollout_ring): with every
        policy internal a step needs nothing from the host (env_utils.py:45-52 passes None until the episode is over),
        so step_lookahead() hands out slot t of the ring and launches the next k steps when it runs dry -- the fused
        n-step kernel never waits for the slowest workgroup of a step (7.7 instead of 14.9 us per step at 4096 x 10).
        Results are those of k single launches, bit for bit.  Whatever needs the state of the step last handed out --
        reading `state`, a reset, an external action, a parameter change, the episode statistics -- goes through sync(),
        which rewinds (restore the snapshot taken before the launch, re-run the steps already handed out).
        fresh: every refill writes into a NEWLY allocated ring (what step_lookahead returned stays valid and belongs to
        the caller); False: one persistent ring, a slot is overwritten k steps later.  k = 0: off.
        adaptive: k is the LONGEST ring.  A rewind throws the rest of a ring away, so a caller who looks at the state (or
        acts) every m steps should not pay for k: the first ring is `start` steps long (default min(k, 8): a caller who
        reads the state after its first step has wasted at most 7), after a rewind at slot t the next ring is t steps
        long (at least 1 = one launch per step), and a ring consumed to its end doubles the next one up to k -- after a
        rewind only once TWO rings in a row have been used up, so that a caller who looks at the state after every step
        stays at one launch per step instead of alternating between rings of 1 and 2.
        What step_lookahead() returns are VIEWS of the ring's tensors (slot t of `[n, E, N, W]` ...): with fresh=True they
        are never written again and stay valid for as long as the caller holds them, but holding ONE of them keeps the
        whole ring allocated (n slots); clone a slot that is kept long-term."""
        self.sync()
        k = int(k)
        if k <= 0:
            self._la = None
            return
        cur = k if not adaptive else max(1, min(k, 8 if start is None else int(start)))
        self._la = dict(n=k, cur=cur, len=0, t=0, slots=None, fresh=bool(fresh), adaptive=bool(adaptive), ring=None,
                        snap=torch.empty_like(self._slab), fills=0, rewinds=0, in_kernel={}, prep=None, co_live=None,
                        streak=2)

    def _la_fill(self):
        la = self._la
        if not self.lookahead_ok():
            raise nat.CagpuError("step_lookahead: this batch needs work between two steps (GA3C-CADRL network, stochastic RVO "
                                 "draws, sensor variants or a static map) -- use step()")
        if self._cs.ext_state:   # (CaState.ext_state belongs to the ONE step() call that was given it: see rollout())
            self._cs.ext_state, self._ext_state = None, None
        if la["adaptive"] and la["slots"] is not None and la["t"] >= la["len"]:   # the last ring was used up: a longer one --
            la["streak"] += 1                                                     # after a rewind, only the second in a row
            if la["streak"] >= 2:
                la["cur"] = min(la["n"], 2 * la["cur"])
        k = la["cur"]
        # Everything a launch needs that does not depend on the moment of the call -- the ring's tensors, the CaOut that names
        # them, the rewind mode, the slot views handed out later -- was prepared behind the PREVIOUS launch (_la_prepare): a
        # caller who synchronises around K steps (bench.py's timed block) has the device idle until this launch is submitted
        prep = la["prep"]
        if prep is None or prep["k"] != k:     # (set_fixture_table / update_params drop a prepared launch)
            prep = self._la_prepare(k)
        la["prep"] = None
        la["ring"] = prep["ring"]
        if not prep["in_kernel"]:
            la["snap"].copy_(self._slab)
        rc = self.lib.cagpu_rollout_ring(self._p_ref, self._cs_ref, prep["co_ref"], None, prep["ar_ref"], k, prep["delta"],
                                         _raw_stream(self._dev_index) if _raw_stream else torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            nat.check(rc)
        la["slots"] = prep["slots"]
        la["co_live"] = prep["co"]      # (keeps the ctypes struct the launch was given alive)
        la["t"], la["len"] = 0, k
        la["fills"] += 1
        if la["fills"] % self.PROBE_EVERY == 1:   # (every refill costs a 20-step block 2.6 us: a second queue for the caller's
            self._fault_probe()                   #  synchronisation to wait on; r06_kernel_geometry.md section 7)
        if la["fresh"]:                 # the next ring (tensors, views, arguments), allocated behind this launch
            la["prep"] = self._la_prepare(min(la["n"], 2 * k) if (la["adaptive"] and la["streak"] >= 1) else k)

    def _la_key(self, k):
        """what decides which kernel a ring call of k steps runs (and with it whether the kernel takes the rewind snapshot)"""
        ar = self._ar
        return (k, self.p.sort_mode, ar is not None and bool(ar.reset_obs), bool(self._cs.next_action),
                0 if ar is None else int(ar.heading_seed), 0 if ar is None else C.addressof(ar))

    def _la_prepare(self, k):
        la, E, N, dev = self._la, self.E, self.N, self.device
        if la["fresh"] or la["ring"] is None or la["ring"][0].shape[0] != k:
            ring = (torch.empty((k, E, N, self.W), dtype=torch.float32, device=dev), torch.empty((k, E, N), dtype=torch.float32, device=dev),
                    torch.empty((k, E, N), dtype=torch.uint8, device=dev), torch.empty((k, E), dtype=torch.uint8, device=dev))
        else:
            ring = la["ring"]           # (fresh=False: ONE persistent ring, a slot is overwritten k steps later)
        obs, rew, done, over = ring
        co = nat.CaOut.from_buffer_copy(self._co)   # (the ring's own CaOut: the workspace of self._co, no actions / orca_vel record)
        co.actions, co.orca_vel = None, None
        co.obs, co.rewards, co.done, co.game_over = obs.data_ptr(), rew.data_ptr(), done.data_ptr(), over.data_ptr()
        ar_ref = None if self._ar is None else C.byref(self._ar)
        # the rewind point = the state BEFORE the k steps: stored by the pipelined n-step kernel itself as it loads its
        # tiles (snapshot_delta: the snapshot slab has the state slab's layout); by one device copy in front of the launch
        # for the other kernels
        key = self._la_key(k)
        in_kernel = la["in_kernel"].get(key)   # (the cache is dropped by set_fixture_table / update_params)
        if in_kernel is None:
            rc = self.lib.cagpu_ring_snapshots(self._p_ref, self._cs_ref, C.byref(co), ar_ref, k)
            if rc < 0:
                nat.check(rc)
            in_kernel = la["in_kernel"][key] = rc == 1
        # (the kernels write 0 / 1 bytes: reinterpreted as bool without a conversion kernel)
        slots = list(zip(obs.unbind(0), rew.unbind(0), done.view(torch.bool).unbind(0), over.view(torch.bool).unbind(0)))
        return dict(k=k, key=key, ring=ring, co=co, co_ref=C.byref(co), ar_ref=ar_ref, in_kernel=in_kernel, slots=slots,
                    delta=(la["snap"].data_ptr() - self._slab.data_ptr()) if in_kernel else 0)

    def step_lookahead(self):
        """one step(None) served from the look-ahead ring -> (obs [E,N,W], rewards [E,N], done [E,N] bool, game_over [E] bool)"""
        la = self._la
        t = la["t"]
        if la["slots"] is None or t >= la["len"]:
            self._la_fill()
            t = 0
        la["t"] = t + 1
        return la["slots"][t]

    def sync(self):
        """Make `state`, `obs`, `rewards`, `done`, `game_over` those of the step LAST HANDED OUT by step_lookahead (a no-op
        without a ring, or when the ring has been consumed to its end): restore the snapshot taken before the ring's launch
        and re-run the t steps already handed out -- the same kernels on the same bits.  The ring is dropped; the next
        step_lookahead() launches a new one."""
        la = self._la
        if la is None or la["slots"] is None:
            return
        t, k = la["t"], la["len"]
        obs, rew, done, over = la["ring"]
        la["slots"] = None
        if t < k:
            la["rewinds"] += 1
            if la["adaptive"]:     # the caller came back after t steps: that is how far the next ring looks ahead
                la["cur"] = max(1, t)
                la["streak"] = 0
        if not la["fresh"]:
            la["ring"] = None      # (the current outputs below live in it: the next fill must not overwrite them)
        if t > 0:                  # the outputs of the last step handed out are the simulator's current outputs
            own = (lambda x: x.clone()) if la["fresh"] else (lambda x: x)   # (a fresh ring's slots belong to the caller)
            self._obs, self._rewards, self._done, self._game_over = own(obs[t - 1]), own(rew[t - 1]), own(done[t - 1]), own(over[t - 1])
            co = self._co
            co.obs, co.rewards, co.done, co.game_over = (self._obs.data_ptr(), self._rewards.data_ptr(), self._done.data_ptr(),
                                                         self._game_over.data_ptr())
        if t < k:
            self._slab.copy_(la["snap"])
            if t > 0:              # (rewrites slot t - 1 with the values it already holds)
                nat.check(self.lib.cagpu_rollout(C.byref(self.p), C.byref(self._cs), C.byref(self._co), None,
                                                 None if self._ar is None else C.byref(self._ar), t, self._stream()))

    # ---------------------------------------------------------------- statistics
    def _fault_probe(self):
        """The device's fault word on the product path, without a synchronisation: a 4-byte copy into pinned host memory
        is queued behind the launch just submitted (cagpu_device_faults_async); the word an EARLIER probe brought back is
        looked at here once its copy has landed.  A raised bit (a hand-over poll of the pipelined kernel ran out, a
        GA3C-CADRL operand left the fp16 range) raises CagpuError through check_faults()."""
        if torch.cuda.is_current_stream_capturing():
            return                   # (a step captured into a HIP graph: events and side streams have no place in the capture)
        fp = self._fault
        if fp is None:
            # the copy runs on a stream of its own: the word is a device global that kernels OR bits into, so the read needs
            # no ordering with the launches -- and on the compute stream a 4-byte device-to-host copy behind every ring launch
            # would sit between the kernel's end and the caller's synchronisation (~5 us of a 200 us block)
            fp = self._fault = dict(buf=torch.zeros((1,), dtype=torch.int32).pin_memory(), ev=torch.cuda.Event(), busy=False, probes=0,
                                    stream=torch.cuda.Stream(device=self.device))
            fp["h"] = (C.c_void_p(fp["buf"].data_ptr()), C.c_void_p(fp["stream"].cuda_stream))
        if fp["busy"]:
            if not fp["ev"].query():
                return               # (still in flight: looked at by a later call)
            fp["busy"] = False
            if int(fp["buf"][0]) != 0:
                self.check_faults()  # (synchronising read + clear; raises)
        with torch.cuda.device(self.device):     # (the word is the CURRENT device's symbol: a process that drives several GPUs)
            rc = self.lib.cagpu_device_faults_async(*fp["h"])
            if rc != 0:
                nat.check(rc)
            fp["ev"].record(fp["stream"])
        fp["busy"] = True
        fp["probes"] += 1

    def check_faults(self):
        """Raise if a step kernel flagged a fault on this device since the last check (cagpu_device_faults: a bounded
        hand-over poll of the pipelined kernel ran out -- the state may be wrong).  Synchronises the device."""
        with torch.cuda.device(self.device):
            f = nat.device_faults(clear=True)
        if f:
            raise nat.CagpuError("device fault word 0x%x:%s%s the simulator state is not to be trusted" % (
                f, " a hand-over inside the pipelined step kernel timed out;" if f & 1 else "",
                " a GA3C-CADRL operand left the fp16 range of the network kernel's two-plane split (|x| >= 65504);" if f & 2 else ""))

    def episode_stats(self, check=True):
        """Per-shard episode counters: float64 [8] (see STAT_NAMES), reduced on the device.  A reporting point: the
        device's fault word is checked here (one small synchronising read)."""
        self.sync()
        if check:
            self.check_faults()
        return self._state["env_stats"].sum(dim=0)


def orca(pos, vel, pref, radius, max_speed, collab=0.5, time_horizon=5.0, time_step=0.1, max_neighbors=None,
         neighbor_dist=math.inf):
    """Batched replacement of rvo2.PyRVOSimulator.doStep() (RVOPolicy.py:93): float32 device tensors
    pos/vel/pref [E,N,2], radius/max_speed [E,N] -> new velocities [E,N,2]."""
    assert pos.is_cuda and pos.dtype == torch.float32
    E, N = pos.shape[:2]
    ts = [t.contiguous() for t in (pos, vel, pref, radius, max_speed)]
    out = torch.empty((E, N, 2), dtype=torch.float32, device=pos.device)
    st = C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
    nat.check(nat.lib().cagpu_orca(E, N, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(),
                                   ts[4].data_ptr(), collab, time_horizon, time_step,
                                   N if max_neighbors is None else max_neighbors, neighbor_dist, out.data_ptr(), st))
    return out
