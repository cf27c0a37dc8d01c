import numpy as np

from gym_collision_avoidance_amd import _native as nat
from gym_collision_avoidance_amd.envs import Config
from .InternalPolicy import InternalPolicy


def _wrap(a):  # util.py:141-146
    while a >= np.pi:
        a -= 2 * np.pi
    while a < -np.pi:
        a += 2 * np.pi
    return a


class RVOPolicy(InternalPolicy):
    """ORCA / RVO2 policy (reference policies/RVOPolicy.py).  The reference keeps one private rvo2 simulator per
    agent and runs a full doStep() for every agent every step; here the half-plane construction and the incremental
    linear program run inside the fused HIP kernel for all agents of all envs at once (csrc/cagpu.hip, csrc/cagpu_pipe.inc;
    stand-alone entry point `cagpu_orca`).  Parameters follow RVOPolicy.py:13-28: timeStep = Config.DT, neighborDist =
    SENSING_HORIZON, maxNeighbors = MAX_NUM_AGENTS_IN_ENVIRONMENT, timeHorizon = RVO_TIME_HORIZON, per-agent
    radius * 1.05 and maxSpeed = pref_speed; the pi/6 turn clip of :109-111 is applied in the kernel.

    `find_next_action(obs, agents, i)` (InternalPolicy.py:12-23) also works on the HOST, for code written against the
    plugin API: it sends the agents' bodies through `cagpu_orca` (rvo2's doStep on the device, bit-identical to the step
    kernel's ORCA phases) and applies RVOPolicy.py:96-122 in numpy -- the same float32 / float64 mix, so the action equals
    the one the step kernel takes.  The reference's stochastic branches -- `heading_noise` (np.random.normal per query,
    :118-119) and a negative RVO_COLLAB_COEFF (np.random.choice every RVO_ANTI_COLLAB_T seconds, :77-88) -- draw from numpy's
    global stream per agent and per step; an agent whose policy has one of them switched on is queried through THIS host
    path (single-env mode; `needs_host`), with the reference's own np.random calls.  `has_fixed_speed` reads
    `self.max_speed`, which RVOPolicy never defines (RVOPolicy.py:114-115 raises AttributeError when switched on): there
    is no behaviour to mirror."""
    kernel_id = nat.POL_RVO

    def __init__(self):
        InternalPolicy.__init__(self, str="RVO")
        self.dt = Config.DT
        self.has_fixed_speed = False   # (see the class docstring)
        self.heading_noise = False
        self.max_delta_heading = np.pi / 6
        self.use_non_coop_policy = True

    @property
    def needs_host(self):
        """True: the stochastic branches are on, the env queries this agent on the host instead of inside the kernel"""
        return bool(self.heading_noise) or Config.RVO_COLLAB_COEFF < 0

    def find_next_action(self, obs, agents, i):
        import torch
        from gym_collision_avoidance_amd import core
        f32 = np.float32
        pos = np.array([a.pos_global_frame for a in agents], dtype=np.float64)
        vel = np.array([a.vel_global_frame for a in agents], dtype=np.float64)
        goal = np.array([a.goal_global_frame for a in agents], dtype=np.float64)
        ps = np.array([a.pref_speed for a in agents], dtype=np.float64)
        rad = np.array([a.radius for a in agents], dtype=np.float64)
        pref = goal - pos                                                    # RVOPolicy.py:66-67
        pref = (ps / np.sqrt((pref * pref).sum(axis=1)))[:, None] * pref
        collab = Config.RVO_COLLAB_COEFF
        if collab < 0:  # RVOPolicy.py:77-88: every RVO_ANTI_COLLAB_T seconds choose between non-cooperative and adversarial
            t, T = agents[i].t, Config.RVO_ANTI_COLLAB_T
            if round(t % T, 3) < Config.DT or round(T - t % T, 3) < Config.DT:
                self.use_non_coop_policy = bool(np.random.choice([True, False], p=[1 - abs(collab), abs(collab)]))
            if self.use_non_coop_policy:
                collab = 0.0
        env = getattr(agents[i], "_env", None)
        dev = torch.device(env.device if env is not None else "cuda:0")
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=f32)[None]).to(dev)
        v = core.orca(t(pos), t(vel), t(pref), t((1 + 5e-2) * rad), t(ps), collab=float(collab),
                      time_horizon=float(Config.RVO_TIME_HORIZON), time_step=float(self.dt),
                      max_neighbors=int(Config.MAX_NUM_AGENTS_IN_ENVIRONMENT),
                      neighbor_dist=float(Config.SENSING_HORIZON))[0, i].cpu().numpy()
        new_pos = pos[i].astype(f32) + v * f32(self.dt)                      # rvo2 Agent::update, C float
        delta = new_pos.astype(np.float64) - pos[i]                          # :97
        new_heading = np.arctan2(delta[1], delta[0]) % (2 * np.pi)           # :100-102
        delta_heading = _wrap(new_heading - agents[i].heading_global_frame)  # :103
        speed = 1 / self.dt * np.linalg.norm(delta)                          # :106
        if abs(delta_heading) > self.max_delta_heading:                      # :109-111
            delta_heading = np.sign(delta_heading) * self.max_delta_heading
            speed = 0.
        if self.heading_noise:                                               # :118-119
            delta_heading = delta_heading + np.random.normal(0, 0.5)
        return np.array([speed, delta_heading])
