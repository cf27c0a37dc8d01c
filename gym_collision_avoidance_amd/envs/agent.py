"""Agent: the reference's per-agent object (gym_collision_avoidance/envs/agent.py) as a thin VIEW.

The reference Agent owns its state as Python attributes and mutates it in `take_action`.  Here the state of every
agent of every env lives in device tensors (core.BatchedSim) and is advanced by the HIP step kernel; an Agent keeps
  * its constructor arguments (the initial condition the env uploads at `env.reset()`), and
  * a binding (env, env index, slot) through which the reference's attribute names read the CURRENT values from a
    host snapshot the env refreshes lazily (one device->host copy per step, only if somebody looks).
So `agent.pos_global_frame`, `agent.is_at_goal`, `agent.t` ... keep working for evaluation scripts
(experiments/src/env_utils.py:56-87) without putting Python objects on the hot path.
"""
import math

import numpy as np

from gym_collision_avoidance_amd import _native as nat
from gym_collision_avoidance_amd.envs import Config


def wrap(angle):
    """[-pi, pi) like the reference's util.wrap (util.py:141-146)."""
    while angle >= np.pi:
        angle -= 2 * np.pi
    while angle < -np.pi:
        angle += 2 * np.pi
    return angle


class Agent(object):
    """Same constructor as the reference (agent.py:29-30): policy / dynamics_model / sensors are CLASSES."""

    def __init__(self, start_x, start_y, goal_x, goal_y, radius, pref_speed, initial_heading, policy, dynamics_model,
                 sensors, id):
        self.policy = policy()
        self.dynamics_model = dynamics_model(self)
        self.sensors = [sensor() for sensor in sensors]
        self.chosen_action_dict = {}
        self.num_actions_to_store = 2
        self.action_dim = 2
        self.id = id
        self.near_goal_threshold = Config.NEAR_GOAL_THRESHOLD
        self.dt_nominal = Config.DT
        self.min_x, self.max_x, self.min_y, self.max_y = -20.0, 20.0, -20.0, 20.0
        self.t_offset = None
        self.global_state_dim = 11
        self.ego_state_dim = 3
        self.max_heading_change = np.pi / 3   # overwritten by the env (collision_avoidance_env.py:364-367)
        self.max_speed = 1.0
        self._env, self._e, self._a = None, 0, None
        self._init = {}
        self._history = []
        self.reset(px=start_x, py=start_y, gx=goal_x, gy=goal_y, pref_speed=pref_speed, radius=radius,
                   heading=initial_heading)

    # ------------------------------------------------------------------ initial condition (agent.py:59-138)
    def reset(self, px=None, py=None, gx=None, gy=None, pref_speed=None, radius=None, heading=None):
        """Record a new initial condition; it is uploaded by the next `env.reset()` (agent state lives on the
        device).  Unspecified fields keep their previous value, like the reference."""
        ini = self._init
        if px is not None and py is not None:
            ini["px"], ini["py"] = float(px), float(py)
        if gx is not None and gy is not None:
            ini["gx"], ini["gy"] = float(gx), float(gy)
        if radius is not None:
            ini["radius"] = float(radius)
        if pref_speed is not None:
            ini["pref_speed"] = float(pref_speed)
        ini["heading"] = (math.atan2(ini["gy"] - ini["py"], ini["gx"] - ini["px"]) if heading is None
                          else float(heading))
        self._unbind()

    def _case_row(self):
        i = self._init
        return [i["px"], i["py"], i["gx"], i["gy"], i["pref_speed"], i["radius"]], i["heading"]

    def _bind(self, env, e, a):
        self._env, self._e, self._a = env, e, a
        self._history = []

    def _unbind(self):
        self._env, self._a = None, None

    # ------------------------------------------------------------------ state access
    def _s(self, name):
        """current value of a state field: from the bound env's host snapshot, else from the initial condition"""
        if self._env is not None:
            return self._env._snapshot()[name][self._e, self._a]
        i = self._init
        slt = (math.hypot(i["px"] - i["gx"], i["py"] - i["gy"]) - self.near_goal_threshold) / i["pref_speed"]
        unbound = {"pos_x": i["px"], "pos_y": i["py"], "goal_x": i["gx"], "goal_y": i["gy"], "vel_x": 0.0,
                   "vel_y": 0.0, "heading": i["heading"], "radius": i["radius"], "pref_speed": i["pref_speed"],
                   "t": 0.0, "slt": slt, "time_remaining": max(Config.MAX_TIME_RATIO * slt, self.dt_nominal),
                   "flags": 0, "step_num": 0, "ep_reward": 0.0, "turning_dir": 0.0, "last_action": np.zeros(2, np.float32)}
        return unbound[name]

    def _flag(self, bit):
        return bool(int(self._s("flags")) & bit)

    pos_global_frame = property(lambda self: np.array([self._s("pos_x"), self._s("pos_y")], dtype="float64"))
    vel_global_frame = property(lambda self: np.array([self._s("vel_x"), self._s("vel_y")], dtype="float64"))
    goal_global_frame = property(lambda self: np.array([self._s("goal_x"), self._s("goal_y")], dtype="float64"))
    heading_global_frame = property(lambda self: float(self._s("heading")))
    radius = property(lambda self: float(self._s("radius")))
    pref_speed = property(lambda self: float(self._s("pref_speed")))
    t = property(lambda self: float(self._s("t")))
    time_remaining_to_reach_goal = property(lambda self: float(self._s("time_remaining")))
    straight_line_time_to_reach_goal = property(lambda self: float(self._s("slt")))
    step_num = property(lambda self: int(self._s("step_num")))
    turning_dir = property(lambda self: float(self._s("turning_dir")))  # (agent.py:133; UnicycleDynamics.py:41-47)
    is_at_goal = property(lambda self: self._flag(nat.AT_GOAL))
    was_at_goal_already = property(lambda self: self._flag(nat.WAS_AT_GOAL))
    in_collision = property(lambda self: self._flag(nat.IN_COLLISION))
    was_in_collision_already = property(lambda self: self._flag(nat.WAS_IN_COLLISION))
    ran_out_of_time = property(lambda self: self._flag(nat.OUT_OF_TIME))
    is_done = property(lambda self: self._flag(nat.DONE))
    speed_global_frame = property(lambda self: float(self._s("last_action")[0]))
    delta_heading_global_frame = property(lambda self: float(self._s("last_action")[1]))

    @property
    def past_actions(self):
        """(2, 2) like the reference; only the most recent action is kept on the device, row 1 is zero."""
        out = np.zeros((self.num_actions_to_store, self.action_dim))
        out[0, :] = self._s("last_action")
        return out

    # ego frame (agent.py:329-349, dynamics/Dynamics.py:24-41), recomputed from the state on demand
    def get_ref(self):
        d = self.goal_global_frame - self.pos_global_frame
        dist = math.sqrt(d[0] ** 2 + d[1] ** 2)
        ref_prll = d / dist if dist > 1e-8 else d
        return ref_prll, np.array([-ref_prll[1], ref_prll[0]])

    ref_prll = property(lambda self: self.get_ref()[0])
    ref_orth = property(lambda self: self.get_ref()[1])

    @property
    def dist_to_goal(self):
        d = self.goal_global_frame - self.pos_global_frame
        return math.sqrt(d[0] ** 2 + d[1] ** 2)

    @property
    def heading_ego_frame(self):
        p = self.ref_prll
        return wrap(self.heading_global_frame - math.atan2(p[1], p[0]))

    @property
    def vel_ego_frame(self):
        v = self.vel_global_frame
        s = math.sqrt(v[0] ** 2 + v[1] ** 2)
        h = self.heading_ego_frame
        return np.array([s * math.cos(h), s * math.sin(h)])

    # observations produced by the kernel for this agent
    def _obs_row(self):
        if self._env is None:
            return None
        return self._env._obs_host()[self._e, self._a]

    @property
    def num_other_agents_observed(self):
        row = self._obs_row()
        return 0 if row is None else int(row[1])

    @property
    def other_agent_states(self):
        row = self._obs_row()
        return np.zeros((7,)) if row is None else row[6:13].astype(np.float64)

    @property
    def sensor_data(self):
        row = self._obs_row()
        K = Config.MAX_NUM_OTHER_AGENTS_OBSERVED
        oa = np.zeros((K, 7)) if row is None else row[6:6 + 7 * K].astype(np.float64).reshape(K, 7)
        data = {"other_agents_states": oa}
        if self._env is not None and self._env._sim is not None and self._env._sim.scan is not None:
            data["laserscan"] = self._env._scan_host()[self._e, self._a].astype(np.float64)
        return data

    def get_sensor_data(self, sensor_name):
        return self.sensor_data.get(sensor_name)

    def get_agent_data(self, attribute):
        return getattr(self, attribute)

    def get_agent_data_equiv(self, attribute, value):
        obj = self
        for part in attribute.split("."):
            obj = getattr(obj, part)
        return obj == value

    def get_observation_dict(self, agents=None):
        """{state: np.array} for Config.STATES_IN_OBS (agent.py:323-327)."""
        getters = {"is_learning": lambda: self.policy.str == "learning",
                   "num_other_agents": lambda: self.num_other_agents_observed,
                   "dist_to_goal": lambda: self.dist_to_goal, "heading_ego_frame": lambda: self.heading_ego_frame,
                   "pref_speed": lambda: self.pref_speed, "radius": lambda: self.radius,
                   "other_agent_states": lambda: self.other_agent_states,
                   "other_agents_states": lambda: self.get_sensor_data("other_agents_states"),
                   "laserscan": lambda: self.get_sensor_data("laserscan")}
        return {s: np.array(getters[s]()) for s in Config.STATES_IN_OBS}

    def sense(self, agents, agent_index, top_down_map):
        """Kept for API compatibility (agent.py:243-255): the sensors already ran in the kernel."""
        return None

    def to_vector(self):
        p, g, v = self.pos_global_frame, self.goal_global_frame, self.vel_global_frame
        global_state = np.array([self.t, p[0], p[1], g[0], g[1], self.radius, self.pref_speed, v[0], v[1],
                                 self.speed_global_frame, self.heading_global_frame])
        return global_state, np.array([self.t, self.dist_to_goal, self.heading_ego_frame])

    @property
    def global_state_history(self):
        """[step, 11] trajectory log (agent.py:257-289), recorded by the env when Config.STORE_HISTORY (host side,
        single-env mode)."""
        return np.array(self._history).reshape(-1, self.global_state_dim)

    def set_state(self, px, py, vx=None, vy=None, heading=None):
        """ExternalDynamics support (agent.py:155-190): overwrite pos / vel / heading of this agent on the device."""
        if self._env is None:
            raise RuntimeError("set_state needs an agent bound to a reset env")
        if vx is None or vy is None:
            if self.step_num == 0:
                vx, vy = 0.0, 0.0
            else:
                old = self.pos_global_frame
                vx, vy = (px - old[0]) / self.dt_nominal, (py - old[1]) / self.dt_nominal
        if heading is None:
            heading = math.atan2(vy, vx)
        self._env._write_agent(self._e, self._a, pos_x=px, pos_y=py, vel_x=vx, vel_y=vy, heading=heading)

    def take_action(self, action, dt):
        raise RuntimeError("agents move inside the HIP step kernel (env.step); there is no per-agent host step")

    def print_agent_info(self):
        print("----------\nGlobal Frame:\n(px,py):", self.pos_global_frame, "\n(vx,vy):", self.vel_global_frame,
              "\nspeed:", self.speed_global_frame, "\nheading:", self.heading_global_frame, "\nBody Frame:\n(vx,vy):",
              self.vel_ego_frame, "\nheading:", self.heading_ego_frame, "\n----------")

    def __deepcopy__(self, memo):
        """A frozen snapshot of the current values (the reference copies everything but the policy,
        agent.py:141-148); used for env.prev_episode_agents."""
        snap = AgentSnapshot()
        for name in ("pos_global_frame", "vel_global_frame", "goal_global_frame", "heading_global_frame", "radius",
                     "pref_speed", "t", "time_remaining_to_reach_goal", "straight_line_time_to_reach_goal",
                     "step_num", "turning_dir", "is_at_goal", "was_at_goal_already", "in_collision", "was_in_collision_already",
                     "ran_out_of_time", "is_done", "dist_to_goal", "heading_ego_frame", "id",
                     "global_state_history"):
            setattr(snap, name, getattr(self, name))
        snap.policy_str = self.policy.str
        return snap


class AgentSnapshot(object):
    """Plain attribute bag produced by copy.deepcopy(agent)."""
