class Sensor(object):
    """Base class (reference sensors/Sensor.py): `name`, `sense(agents, agent_index, top_down_map)`, `set_args`."""
    name = None

    def __init__(self):
        pass

    def sense(self, agents, agent_index, top_down_map):
        raise NotImplementedError

    def set_args(self, args):
        for arg, value in args.items():
            setattr(self, arg, value)
