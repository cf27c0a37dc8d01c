class Sensor(object):
    """Plugin base class with the reference's surface (sensors/Sensor.py): a `name` under which the measurement is
    stored in `agent.sensor_data`, `sense(agents, agent_index, top_down_map)` and `set_args({attribute: value})`.

    The built-in sensors are evaluated for every agent of every env inside the HIP kernels; their `sense` returns the
    row the kernel produced.  `kernel_args` lists the attributes the env forwards to the kernel parameters."""
    name = None
    kernel_args = ()

    def sense(self, agents, agent_index, top_down_map):
        raise NotImplementedError("%s does not implement sense()" % type(self).__name__)

    def set_args(self, args):
        """Overwrite attributes from a dict (env_utils.py policies[...]['sensor_args'])."""
        for key in args:
            setattr(self, key, args[key])
