class Dynamics(object):
    """Base class (reference dynamics/Dynamics.py).  The state update itself (`step`, `update_ego_frame`) happens
    in the HIP kernel; `kernel_id` selects which one."""
    kernel_id = None

    def __init__(self, agent):
        self.agent = agent

    def step(self, action, dt):
        raise RuntimeError("dynamics run inside the HIP step kernel (csrc/cagpu.hip), not per agent on the host")

    def update_ego_frame(self):
        """No-op on the host: dist_to_goal / heading_ego_frame / ref_prll are produced by the kernel and read
        through the Agent view."""
