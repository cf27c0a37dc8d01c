"""Import-only stand-in for tensorflow (GA3C_CADRL/network.py imports tensorflow.compat.v1)."""
