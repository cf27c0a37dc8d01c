from gym_collision_avoidance_amd import _native as nat
from .Dynamics import Dynamics


class UnicycleDynamics(Dynamics):
    """heading' = wrap(heading + a[1]); pos += a[0] * (cos, sin)(heading') * dt (reference
    dynamics/UnicycleDynamics.py:14-47)."""
    kernel_id = nat.DYN_UNICYCLE

    def step(self, action, dt):
        """Host-callable like the reference's (UnicycleDynamics.py:14-47): turn to wrap(action[1] + heading), drive
        action[0] for dt, update turning_dir -- written to the agent's device state."""
        from gym_collision_avoidance_amd.envs.agent import wrap
        self._host_unicycle(action[0], wrap(action[1] + self.agent.heading_global_frame), dt, True)
