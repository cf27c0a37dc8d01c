from .Sensor import Sensor
from .OtherAgentsStatesSensor import OtherAgentsStatesSensor
from .LaserScanSensor import LaserScanSensor
