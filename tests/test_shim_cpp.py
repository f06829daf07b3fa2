"""The C++ host shim classes (pointslot_amd/host/) compile with plain g++ against the C-ABI — no OpenCV, no Eigen —
and (on a GPU box) reproduce the oracle through an ORB_SLAM2-shaped caller."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_smoke")


def _build():
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "pointslot_amd", "host"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "shim_smoke.cpp"), "-o", EXE, "-L", os.path.join(ROOT, "pointslot_amd"),
           "-lpointslot_hip", "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"), "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_shim_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_shim_runs_and_matches_oracle(tmp_path):
    from oracle_lib import OracleORB, OracleCvORB, KEYPOINT_DTYPE
    from pointslot_amd import synth
    _build()          # always: a binary that travelled with the snapshot may predate the headers
    left, _ = synth.stereo_pair(w=800, h=300)
    raw = tmp_path / "img.raw"
    left.tofile(raw)
    out = subprocess.run([EXE, str(raw), "800", "300", str(tmp_path / "o")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim smoke ok" in out.stdout
    kps = np.fromfile(tmp_path / "o.kps", KEYPOINT_DTYPE)
    desc = np.fromfile(tmp_path / "o.desc", np.uint8).reshape(-1, 32)
    ko, do = OracleORB(1000).run(left)
    assert np.array_equal(kps.view(np.uint8), ko.view(np.uint8))
    assert np.array_equal(desc, do)
    # OpencvORBDetector(im, ObjMask, kp, descriptor) == cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute (the CPU restatement of it)
    okps = np.fromfile(tmp_path / "o.okps", KEYPOINT_DTYPE)
    odesc = np.fromfile(tmp_path / "o.odesc", np.uint8).reshape(-1, 32)
    mask = np.zeros_like(left)
    mask[300 // 6 + 1:, 800 // 8 + 1:800 // 2] = 255
    ko, do = OracleCvORB().run(left, mask)
    assert len(okps) == len(ko) and len(ko) > 100
    assert np.array_equal(okps.view(np.uint8), ko.view(np.uint8)) and np.array_equal(odesc, do)
