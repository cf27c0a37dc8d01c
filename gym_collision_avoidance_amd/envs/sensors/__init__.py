from .Sensor import Sensor
from .OtherAgentsStatesSensor import OtherAgentsStatesSensor
