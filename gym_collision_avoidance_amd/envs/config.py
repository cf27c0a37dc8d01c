"""Config classes with the reference's attribute names and defaults (gym_collision_avoidance/envs/config.py).
Only the attributes the simulator reads are meaningful here; display / plotting switches are kept so that config
files written for the reference keep loading.  As in the reference, attributes that size the observation
(MAX_NUM_AGENTS_IN_ENVIRONMENT, MAX_NUM_OTHER_AGENTS_OBSERVED, STATES_IN_OBS, USE_STATIC_MAP) must be set BEFORE
Config.__init__ runs (config.py:64-70)."""
import numpy as np


def _state(size, bounds, attr, mean=None, std=None):
    d = {"dtype": np.float32, "size": size, "bounds": bounds, "attr": attr}
    if mean is not None:
        d["mean"] = mean
    if std is not None:
        d["std"] = std
    return d


class Config(object):
    def __init__(self):
        f32 = lambda *v: np.array(v, dtype=np.float32)
        # general
        self.COLLISION_AVOIDANCE = True
        self.continuous, self.discrete = range(2)
        self.ACTION_SPACE_TYPE = self.continuous
        # display (host-side plotting is out of scope; kept for config compatibility)
        self.ANIMATE_EPISODES = self.SHOW_EPISODE_PLOTS = self.SAVE_EPISODE_PLOTS = False
        self._default("PLOT_CIRCLES_ALONG_TRAJ", True)
        self.ANIMATION_PERIOD_STEPS = 5
        self.PLT_LIMITS = None
        self.PLT_FIG_SIZE = (10, 8)
        self._default("USE_STATIC_MAP", False)
        # train / play / evaluate
        self.TRAIN_MODE, self.PLAY_MODE, self.EVALUATE_MODE = True, False, False
        # rewards
        self.REWARD_AT_GOAL = 1.0
        self.REWARD_COLLISION_WITH_AGENT = -0.25
        self.REWARD_COLLISION_WITH_WALL = -0.25
        self.REWARD_GETTING_CLOSE = -0.1
        self.REWARD_ENTERED_NORM_ZONE = -0.05
        self.REWARD_TIME_STEP = 0.0
        self.REWARD_WIGGLY_BEHAVIOR = 0.0
        self.WIGGLY_BEHAVIOR_THRESHOLD = np.inf
        self.COLLISION_DIST = 0.0
        self.GETTING_CLOSE_RANGE = 0.2
        self.SOCIAL_NORMS = "none"
        # simulation
        self.DT = 0.2
        self.NEAR_GOAL_THRESHOLD = 0.2
        self.MAX_TIME_RATIO = 2.0
        # test cases
        self.TEST_CASE_FN = "get_testcase_random"
        self.TEST_CASE_ARGS = {
            "policy_to_ensure": "learning_ga3c",
            "policies": ["noncoop", "learning_ga3c", "static"],
            "policy_distr": [0.05, 0.9, 0.05],
            "speed_bnds": [0.5, 2.0],
            "radius_bnds": [0.2, 0.8],
            "side_length": [{"num_agents": [0, 5], "side_length": [4, 5]},
                            {"num_agents": [5, np.inf], "side_length": [6, 8]}],
        }
        self._default("MAX_NUM_AGENTS_IN_ENVIRONMENT", 4)
        self._default("MAX_NUM_AGENTS_TO_SIM", 4)
        self.MAX_NUM_OTHER_AGENTS_IN_ENVIRONMENT = self.MAX_NUM_AGENTS_IN_ENVIRONMENT - 1
        self._default("MAX_NUM_OTHER_AGENTS_OBSERVED", self.MAX_NUM_AGENTS_IN_ENVIRONMENT - 1)
        self.PLOT_EVERY_N_EPISODES = 100
        # sensors
        self.SENSING_HORIZON = np.inf
        self.LASERSCAN_LENGTH = 512
        self.LASERSCAN_NUM_PAST = 3
        self.NUM_STEPS_IN_OBS_HISTORY = 1
        self.NUM_PAST_ACTIONS_IN_STATE = 0
        # RVO agents
        self.RVO_TIME_HORIZON = 5.0
        self.RVO_COLLAB_COEFF = 0.5
        self.RVO_ANTI_COLLAB_T = 1.0
        # storage
        self.STORE_HISTORY = True
        # observation vector
        self.TRAIN_SINGLE_AGENT = False
        K = self.MAX_NUM_OTHER_AGENTS_OBSERVED
        oth_std, oth_mean = f32(5, 5, 1, 1, 1, 5, 1), f32(0, 0, 0, 0, 0.5, 0, 1)
        scan = (self.LASERSCAN_NUM_PAST, self.LASERSCAN_LENGTH)
        inf = np.inf
        self.STATE_INFO_DICT = {
            "dist_to_goal": _state(1, [-inf, inf], 'get_agent_data("dist_to_goal")', f32(0.), f32(5.)),
            "radius": _state(1, [0, inf], 'get_agent_data("radius")', f32(0.5), f32(1.0)),
            "heading_ego_frame": _state(1, [-np.pi, np.pi], 'get_agent_data("heading_ego_frame")', f32(0.), f32(3.14)),
            "pref_speed": _state(1, [0, inf], 'get_agent_data("pref_speed")', f32(1.0), f32(1.0)),
            "num_other_agents": _state(1, [0, inf], 'get_agent_data("num_other_agents_observed")', f32(1.0), f32(1.0)),
            "other_agent_states": _state(7, [-inf, inf], 'get_agent_data("other_agent_states")', oth_mean, oth_std),
            "other_agents_states": _state((K, 7), [-inf, inf], 'get_sensor_data("other_agents_states")',
                                          np.tile(oth_mean, (K, 1)), np.tile(oth_std, (K, 1))),
            "laserscan": _state(scan, [0., 6.], 'get_sensor_data("laserscan")',
                                5. * np.ones(scan, np.float32), 5. * np.ones(scan, np.float32)),
            "is_learning": _state(1, [0., 1.], 'get_agent_data_equiv("policy.str", "learning")'),
            "other_agents_states_encoded": _state(100., [0., 1.], 'get_sensor_data("other_agents_states_encoded")'),
        }
        self.setup_obs()
        self.AGENT_SORTING_METHOD = "closest_first"

    def _default(self, name, value):
        if not hasattr(self, name):
            setattr(self, name, value)

    def setup_obs(self):
        self._default("STATES_IN_OBS", ["is_learning", "num_other_agents", "dist_to_goal", "heading_ego_frame",
                                        "pref_speed", "radius", "other_agents_states"])
        self._default("STATES_NOT_USED_IN_POLICY", ["is_learning"])
        self.MEAN_OBS, self.STD_OBS = {}, {}
        for s in self.STATES_IN_OBS:
            info = self.STATE_INFO_DICT[s]
            if "mean" in info:
                self.MEAN_OBS[s] = info["mean"]
            if "std" in info:
                self.STD_OBS[s] = info["std"]


class EvaluateConfig(Config):
    def __init__(self):
        self._default("MAX_NUM_AGENTS_IN_ENVIRONMENT", 19)
        Config.__init__(self)
        self.EVALUATE_MODE, self.TRAIN_MODE = True, False
        self.DT = 0.1
        self.MAX_TIME_RATIO = 8.0


class Example(EvaluateConfig):
    def __init__(self):
        EvaluateConfig.__init__(self)
        self.PLOT_CIRCLES_ALONG_TRAJ = True


class SmallTestSuite(EvaluateConfig):
    def __init__(self):
        EvaluateConfig.__init__(self)
        self.NUM_TEST_CASES = 4


class FullTestSuite(EvaluateConfig):
    def __init__(self):
        self.MAX_NUM_OTHER_AGENTS_OBSERVED = 19
        EvaluateConfig.__init__(self)
        self.NUM_TEST_CASES = 4
        self.NUM_AGENTS_TO_TEST = [2, 3, 4]
        self.RECORD_PICKLE_FILES = False
        self.POLICIES_TO_TEST = ["RVO"]
        self.FIXED_RADIUS_AND_VPREF = False
        self.NEAR_GOAL_THRESHOLD = 0.2


class BatchedRVO10(EvaluateConfig):
    """The BASELINE.json metric config: 10 agents per env, K = 9, RVO, EvaluateConfig constants, no history."""
    def __init__(self):
        self.MAX_NUM_AGENTS_IN_ENVIRONMENT = 10
        EvaluateConfig.__init__(self)
        self.STORE_HISTORY = False
