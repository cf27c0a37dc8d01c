"""Import-only stand-in for imageio (the reference imports it in Map.py / visualize.py)."""


def imread(*a, **k):
    raise RuntimeError("imageio stub: no image IO in the oracle harness")
