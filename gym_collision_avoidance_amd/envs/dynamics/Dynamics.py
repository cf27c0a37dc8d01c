class Dynamics(object):
    """Base class (reference dynamics/Dynamics.py).  The state update itself (`step`, `update_ego_frame`) happens
    in the HIP kernel; `kernel_id` selects which one."""
    kernel_id = None

    def __init__(self, agent):
        self.agent = agent

    def step(self, action, dt):
        """The built-in models are host-callable like the reference's (they integrate ONE agent on the host and write
        the result to its device state: see UnicycleDynamics.step); inside env.step() the same arithmetic runs in the HIP
        kernel for every agent of the batch.  A custom subclass cannot be run by the kernel."""
        raise RuntimeError("this dynamics model has no kernel counterpart: env.step() integrates UnicycleDynamics / "
                           "UnicycleDynamicsMaxTurnRate / ExternalDynamics (csrc/cagpu.hip); move a custom model's agent "
                           "with ExternalDynamics + Agent.set_state")

    def _host_unicycle(self, speed, selected_heading, dt, update_turning_dir):
        """dynamics/UnicycleDynamics.py:30-47 for self.agent, through the Agent view (device state of a bound agent)"""
        import numpy as np
        a = self.agent
        if a._env is None:
            raise RuntimeError("Dynamics.step needs an agent bound to a reset env (env.set_agents + env.reset)")
        pos = a.pos_global_frame
        px = pos[0] + speed * np.cos(selected_heading) * dt
        py = pos[1] + speed * np.sin(selected_heading) * dt
        fields = dict(pos_x=px, pos_y=py, vel_x=speed * np.cos(selected_heading), vel_y=speed * np.sin(selected_heading),
                      heading=selected_heading)
        if update_turning_dir:
            td = a.turning_dir
            if abs(td) < 1e-5:
                td = 0.11 * np.sign(selected_heading)
            elif td * selected_heading < 0:
                td = max(-np.pi, min(np.pi, -td + selected_heading))
            else:
                td = np.sign(td) * max(0.0, abs(td) - 0.1)
            fields["turning_dir"] = td
        a._env._write_agent(a._e, a._a, **fields)

    def update_ego_frame(self):
        """No-op on the host: dist_to_goal / heading_ego_frame / ref_prll are produced by the kernel and read
        through the Agent view."""
