"""oracle/golden_configs.py -- TEST INFRASTRUCTURE.  Config classes handed to the UNMODIFIED reference
through its own GYM_CONFIG_PATH / GYM_CONFIG_CLASS mechanism (gym_collision_avoidance/envs/__init__.py:4-18)
when oracle/gen_golden.py records golden vectors.  They only set attributes the reference already defines
(config.py:3-200); MAX_NUM_AGENTS_IN_ENVIRONMENT must be set BEFORE Config.__init__ (config.py:64-70)."""
import importlib.util
import os

_ref = os.environ.get("CA_REFERENCE_ROOT", "/root/reference")
_spec = importlib.util.spec_from_file_location(
    "_ref_config", os.path.join(_ref, "gym_collision_avoidance", "envs", "config.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
Config = _mod.Config


class _Eval(Config):
    """EvaluateConfig (config.py:193-200) with a chosen agent budget / K / sort order."""
    N_MAX = 10
    K = None
    SORT = "closest_first"

    def __init__(self):
        self.MAX_NUM_AGENTS_IN_ENVIRONMENT = self.N_MAX
        if self.K is not None:
            self.MAX_NUM_OTHER_AGENTS_OBSERVED = self.K
        Config.__init__(self)
        self.EVALUATE_MODE = True
        self.TRAIN_MODE = False
        self.DT = 0.1
        self.MAX_TIME_RATIO = 8.
        self.STORE_HISTORY = False
        self.AGENT_SORTING_METHOD = self.SORT


class Bench10(_Eval):       # the metric config: N=10, K=9, closest_first
    N_MAX = 10


class Swap4(_Eval):         # config 1: 4-agent swap, K=3
    N_MAX = 4


class Small3(_Eval):
    N_MAX = 3


class Clip6(_Eval):         # K < N-1 (clipping) and closest_last ordering
    N_MAX = 6
    K = 3
    SORT = "closest_last"


class Tti6(_Eval):         # time_to_impact ordering, K < N-1
    N_MAX = 6
    K = 4
    SORT = "time_to_impact"


class Pad5(_Eval):          # K > N-1 (zero padding), 5 agents in a 8-slot env
    N_MAX = 5
    K = 7


class Odd6(_Eval):          # every constant the hot path reads moved off its default (finite sensing horizon included)
    N_MAX = 6
    K = 4

    def __init__(self):
        _Eval.__init__(self)
        self.DT = 0.2
        self.MAX_TIME_RATIO = 3.0
        self.SENSING_HORIZON = 4.0
        self.NEAR_GOAL_THRESHOLD = 0.35
        self.GETTING_CLOSE_RANGE = 0.45
        self.REWARD_AT_GOAL = 1.5
        self.REWARD_COLLISION_WITH_AGENT = -0.4
        self.REWARD_TIME_STEP = -0.01
        self.REWARD_WIGGLY_BEHAVIOR = -0.02
        self.WIGGLY_BEHAVIOR_THRESHOLD = 0.15
        self.RVO_TIME_HORIZON = 3.0
        self.RVO_COLLAB_COEFF = 0.35


class Train5(Config):       # training-mode rules: DT=0.2, MAX_TIME_RATIO=2, game over when learners done
    def __init__(self):
        self.MAX_NUM_AGENTS_IN_ENVIRONMENT = 5
        Config.__init__(self)
        self.STORE_HISTORY = False


class Laser4(_Eval):        # static map + LaserScanSensor (config 5 in miniature): 'laserscan' joins the observation
    N_MAX = 4

    def __init__(self):
        self.USE_STATIC_MAP = True
        self.STATES_IN_OBS = ['is_learning', 'num_other_agents', 'dist_to_goal', 'heading_ego_frame', 'pref_speed',
                              'radius', 'other_agents_states', 'laserscan']
        _Eval.__init__(self)
