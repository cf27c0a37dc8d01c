import numpy as np

from .ExternalPolicy import ExternalPolicy


class CARRLPolicy(ExternalPolicy):
    """Discrete-index -> action table of the CARRL wrapper (reference policies/CARRLPolicy.py); the policy itself is
    external to the simulator."""

    def __init__(self):
        ExternalPolicy.__init__(self, str="CARRL")
        n = 11
        self.actions = np.zeros((n, 2))
        self.actions[:, 0] = 1.0
        self.actions[:, 1] = np.linspace(-np.pi / 6, np.pi / 6, n)

    def convert_to_action(self, discrete_action):
        return self.actions[discrete_action, :]
