"""CPU tests of the boundary: libpointslot_hip.so loads without a GPU and exports every function that
include/pointslot_hip.h declares; without a device the create calls fail loudly (no CPU fallback)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pointslot_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from pointslot_amd import _lib
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(_lib.lib, n), "libpointslot_hip.so does not export %s" % n


def test_keypoint_struct_is_cv_keypoint_sized():
    from pointslot_amd.extractor import KEYPOINT_DTYPE
    assert KEYPOINT_DTYPE.itemsize == 28
    assert KEYPOINT_DTYPE.fields["octave"][1] == 20


def test_no_cpu_fallback_without_device():
    import pytest
    from pointslot_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from pointslot_amd.extractor import ORBextractor
    with pytest.raises(_lib.PointslotError) as e:
        ORBextractor(1000, 1.2, 8, 20, 5)
    assert e.value.code == _lib.PS_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pointslot_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % f
