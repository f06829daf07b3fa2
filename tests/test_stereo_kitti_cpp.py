"""examples/stereo_kitti.cpp — the reference's stereo driver shape on the C++ host shim (ORBextractor / ORBmatcher / Optimizer
classes + StereoOdometry.h): compiles with plain g++ against the C-ABI, and on a GPU box tracks a generated sequence in the
reference's on-disk layout to the same trajectory as the Python driver of the same kernels."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "stereo_kitti")


def _build():
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "pointslot_amd", "host"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "stereo_kitti.cpp"), "-o", EXE, "-L", os.path.join(ROOT, "pointslot_amd"),
           "-lpointslot_hip", "-pthread", "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"), "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_stereo_kitti_cpp_compiles_and_links():
    _build()
    assert os.path.exists(EXE)
    out = subprocess.run([EXE], capture_output=True, text=True)
    assert out.returncode == 1 and "Usage" in out.stderr


@pytest.mark.gpu
def test_stereo_kitti_cpp_tracks_like_the_python_driver(tmp_path):
    from pointslot_amd import sequence
    from pointslot_amd.tracker import HipBackend, StereoOdometry
    if not os.path.exists(EXE):
        _build()
    seq = sequence.generate(n_frames=8, seed=4)
    d = str(tmp_path / "0000")
    sequence.write(d, seq, pgm=True)
    out = subprocess.run([EXE, d], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "trajectory saved!" in out.stdout and "LOST" not in out.stdout
    traj = np.loadtxt(os.path.join(d, "CameraTrajectory.txt"))
    assert traj.shape == (8, 12)
    # ground truth of the generator
    assert np.abs(traj[:, 3] - seq["twc"][:, 0, 3]).max() < 0.03 and np.abs(traj[:, [7, 11]]).max() < 0.03
    # the Python driver over the same kernels (host float arithmetic differs in the last ulp: tolerance, not bit-equality)
    be = HipBackend()
    h, w = seq["left"][0].shape
    vo = StereoOdometry(be, seq["K"], seq["bf"], w, h)
    for l, r in zip(seq["left"], seq["right"]):
        vo.track(l, r)
    be.close()
    py = np.array([np.concatenate([t[:3, :3].T, (-(t[:3, :3].T @ t[:3, 3]))[:, None]], 1).reshape(12) for t in vo.trajectory])
    assert np.abs(py - traj).max() < 2e-3, np.abs(py - traj).max()
