"""Nothing in the product may depend on what fresh device memory or LDS happens to contain.  The LDS of every CU is poisoned
before each GPU test (conftest.py); this file repeats a slice of the suite with freshly allocated HBM buffers filled with 0xFF
bytes (NaN as floating point, huge as integers) instead of whatever the driver hands out."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_suite_slice_with_poisoned_device_memory():
    env = dict(os.environ, PS_DEBUG_FILL="255", PS_BA_FILL="255")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-k", "not sweep and not poisoned",
                        os.path.join(ROOT, "tests", "test_opt_gpu.py"), os.path.join(ROOT, "tests", "test_match_gpu.py"),
                        os.path.join(ROOT, "tests", "test_tracker_gpu.py"), os.path.join(ROOT, "tests", "test_orb_gpu.py")],
                       capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
