from gym_collision_avoidance_amd.envs import Config
from .Sensor import Sensor


class OtherAgentsStatesSensor(Sensor):
    """Ego-frame relative states of the closest other agents, distance-sorted, zero-padded to
    (Config.MAX_NUM_OTHER_AGENTS_OBSERVED, 7) (reference sensors/OtherAgentsStatesSensor.py).  Computed for every
    agent of every env inside the HIP step kernel (pair phases P3/P4 of csrc/cagpu.hip; stand-alone entry point
    `cagpu_observe`); `sense` returns this agent's rows from the env's observation tensor."""

    def __init__(self, max_num_other_agents_observed=None, agent_sorting_method=None):
        Sensor.__init__(self)
        self.name = "other_agents_states"
        self.max_num_other_agents_observed = (Config.MAX_NUM_OTHER_AGENTS_OBSERVED
                                              if max_num_other_agents_observed is None
                                              else max_num_other_agents_observed)
        self.agent_sorting_method = (Config.AGENT_SORTING_METHOD if agent_sorting_method is None
                                     else agent_sorting_method)

    def sense(self, agents, agent_index, top_down_map=None):
        return agents[agent_index].get_sensor_data(self.name)
