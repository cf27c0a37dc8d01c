import numpy as np

from gym_collision_avoidance_amd import _native as nat
from gym_collision_avoidance_amd.envs import Config
from .InternalPolicy import InternalPolicy


class RVOPolicy(InternalPolicy):
    """ORCA / RVO2 policy (reference policies/RVOPolicy.py).  The reference keeps one private rvo2 simulator per
    agent and runs a full doStep() for every agent every step; here the half-plane construction and the incremental
    linear program run inside the fused HIP kernel for all agents of all envs at once (csrc/cagpu.hip, stand-alone
    entry point `cagpu_orca`).  Parameters follow RVOPolicy.py:13-28: timeStep = Config.DT, neighborDist =
    SENSING_HORIZON, maxNeighbors = MAX_NUM_AGENTS_IN_ENVIRONMENT, timeHorizon = RVO_TIME_HORIZON, per-agent
    radius * 1.05 and maxSpeed = pref_speed; the pi/6 turn clip of :109-111 is applied in the kernel.
    The reference's optional branches: `has_fixed_speed` reads `self.max_speed`, which RVOPolicy never defines
    (RVOPolicy.py:114-115 raises AttributeError when switched on), so there is no behaviour to mirror; `heading_noise`
    (np.random.normal per query, :118-119) and a negative RVO_COLLAB_COEFF (np.random.choice every RVO_ANTI_COLLAB_T
    seconds, :77-88) draw from numpy's global stream per agent and per step: not reproducible on the device, they
    raise here (both are off in every shipped config)."""
    kernel_id = nat.POL_RVO

    def __init__(self):
        InternalPolicy.__init__(self, str="RVO")
        self.dt = Config.DT
        self.has_fixed_speed = False   # (see the class docstring)
        self.heading_noise = False
        self.max_delta_heading = np.pi / 6
        if Config.RVO_COLLAB_COEFF < 0:
            raise NotImplementedError("anti-collaborative RVO (RVOPolicy.py:77-88) draws from np.random: not ported")

    def find_next_action(self, obs, agents, i):
        raise RuntimeError("RVOPolicy runs inside the HIP step kernel; it has no per-agent host implementation")
