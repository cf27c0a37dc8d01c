from .Dynamics import Dynamics
from .UnicycleDynamics import UnicycleDynamics
from .UnicycleDynamicsMaxTurnRate import UnicycleDynamicsMaxTurnRate
from .ExternalDynamics import ExternalDynamics
