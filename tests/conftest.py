import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _no_device_faults(request):
    """after every GPU test: the device's fault word (include/cagpu.h cagpu_device_faults) must still be zero -- a hand-over
    poll of the pipelined step kernel that ran out would otherwise only show as wrong numbers somewhere"""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from gym_collision_avoidance_amd import _native as nat
    except Exception:  # noqa: BLE001
        return
    assert nat.device_faults(clear=True) == 0, "a step kernel raised the device fault word during this test"
