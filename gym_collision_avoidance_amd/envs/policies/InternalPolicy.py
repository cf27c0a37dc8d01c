from .Policy import Policy


class InternalPolicy(Policy):
    """Computes its action inside the environment (reference policies/InternalPolicy.py)."""

    def __init__(self, str="Internal"):
        Policy.__init__(self, str=str)

    def find_next_action(self, obs, agents, i):
        """-> np.array([speed, delta_heading]) (code order, see UnicycleDynamics.py:27-28).  Built-in subclasses run
        in the HIP kernel and never reach this method; user subclasses implement it (host fallback)."""
        raise NotImplementedError
