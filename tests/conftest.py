import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """LDS keeps the previous kernel's bytes; a kernel that reads a slot it never wrote (or multiplies it by zero) passes or
    fails depending on what ran before.  Every GPU test therefore starts with NaN bit patterns in the LDS of all CUs, which turns
    that class of bug into a deterministic failure."""
    if request.node.get_closest_marker("gpu") is not None:
        from pointslot_amd._lib import poison_lds
        poison_lds(0xFFFFFFFF)
    yield
