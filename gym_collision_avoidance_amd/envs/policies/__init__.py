"""Policy plugins (reference: gym_collision_avoidance/envs/policies/).  A Policy subclass that sets `kernel_id`
is executed inside the fused HIP step kernel (csrc/cagpu.hip); one that does not is queried on the host through the
reference's own `find_next_action(obs, agents, i)` / `external_action_to_action(agent, a)` plugin API and its result
is handed to the kernel as a raw [speed, delta heading] command (the slow, fully general path)."""
from .Policy import Policy
from .InternalPolicy import InternalPolicy
from .ExternalPolicy import ExternalPolicy
from .NonCooperativePolicy import NonCooperativePolicy
from .StaticPolicy import StaticPolicy
from .LearningPolicy import LearningPolicy
from .LearningPolicyGA3C import LearningPolicyGA3C
from .RVOPolicy import RVOPolicy
from .CARRLPolicy import CARRLPolicy
from .GA3CCADRLPolicy import GA3CCADRLPolicy
