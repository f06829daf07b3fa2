"""g2o::ObjectState (SE3 + scale, velocity prediction, cuboid -> bbox; /root/reference/include/g2o_Object.h:30-93,
src/g2o_Object.cc:58-182) and the never-instantiated object edge of row a21 (src/g2o_Object.cc:404-480) as pointslot_amd/host/g2o_Object.h
restates them: known answers from independent numpy / scipy computations, and the a21 stereo edge with Tcw = I against the CPU
checker's a16 edge (EdgeStereoSE3ProjectXYZ) - the reduction DESIGN.md section 1 relies on.  Host code only."""
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "g2o_object_kat")
K4 = (721.5377, 721.5377, 609.5593, 172.854)


@pytest.fixture(scope="module")
def kat():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "pointslot_amd", "host"), EXE + ".cpp", "-o", EXE])

    def run(lines):
        out = subprocess.run([EXE], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
        return [np.array([float(v) for v in l.split()]) for l in out.stdout.splitlines()]
    return run


def _p7(R, t):
    q = Rotation.from_matrix(R).as_quat()          # x y z w
    if q[3] < 0:
        q = -q
    return np.concatenate([t, q])


def _mat(p7):
    M = np.eye(4)
    M[:3, :3] = Rotation.from_quat(p7[3:]).as_matrix()
    M[:3, 3] = p7[:3]
    return M


def _fmt(*parts):
    return " ".join("%.17g" % v for p in parts for v in np.atleast_1d(p))


def test_cuboid_corners_and_bbox(kat):
    rng = np.random.default_rng(20)
    body = {0: np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], float),
            1: np.array([[1, 1, -1, -1, 1, 1, -1, -1], [0, 0, 0, 0, -2, -2, -2, -2], [1, -1, -1, 1, 1, -1, -1, 1]], float)}
    cases, lines = [], []
    for k in range(12):
        Two = _p7(Rotation.from_euler("zyx", rng.uniform(-1, 1, 3)).as_matrix(), rng.uniform(-3, 3, 3) + [0, 0, 12])
        Tcw = _p7(Rotation.from_euler("zyx", rng.uniform(-0.2, 0.2, 3)).as_matrix(), rng.uniform(-1, 1, 3))
        scale = rng.uniform(1, 5, 3)
        centre = k % 2
        cases.append((Two, Tcw, scale, centre))
        lines.append("state " + _fmt(Two, scale) + " %d " % centre + _fmt(Tcw, K4))
    for (Two, Tcw, scale, centre), out in zip(cases, kat(lines)):
        corners = (_mat(Two)[:3, :3] @ np.diag(scale / 2) @ body[centre]) + _mat(Two)[:3, 3:4]
        assert np.allclose(out[:24].reshape(3, 8), corners, atol=1e-12)
        cam = _mat(Tcw)[:3, :3] @ corners + _mat(Tcw)[:3, 3:4]
        u = K4[0] * cam[0] / cam[2] + K4[2]
        v = K4[1] * cam[1] / cam[2] + K4[3]
        rect = np.array([u.min(), v.min(), u.max(), v.max()])
        assert np.allclose(out[24:28], rect, atol=1e-9)
        assert np.allclose(out[28:32], [(rect[0] + rect[2]) / 2, (rect[1] + rect[3]) / 2, rect[2] - rect[0], rect[3] - rect[1]], atol=1e-9)
        assert np.allclose(out[32:36], rect, atol=1e-9)       # transform_from(Tcw) then the from-camera variant: same box
    # a unit cube 10 m in front of an identity camera: the rectangle is symmetric about the principal point
    out, = kat(["state " + _fmt([0, 0, 10, 0, 0, 0, 1], [2, 2, 2]) + " 0 " + _fmt([0, 0, 0, 0, 0, 0, 1], K4)])
    half = K4[0] * 1 / 9
    assert np.allclose(out[24:28], [K4[2] - half, K4[3] - half, K4[2] + half, K4[3] + half], atol=1e-9)


def test_velocity_prediction_and_minimal_vector(kat):
    rng = np.random.default_rng(21)
    cases, lines = [], []
    for _ in range(10):
        T = _p7(Rotation.from_euler("zyx", rng.uniform(-1, 1, 3)).as_matrix(), rng.uniform(-5, 5, 3))
        vel = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-10, 10, 3)])
        dt = float(rng.uniform(0.05, 0.2))
        cases.append((T, vel, dt))
        lines.append("predict " + _fmt(T, vel, dt))
    for (T, vel, dt), out in zip(cases, kat(lines)):
        D = np.eye(4)
        D[:3, :3] = Rotation.from_rotvec(vel[:3] * dt).as_matrix()      # SE3Quat::exp of a pure rotation, then setTranslation(v dt)
        D[:3, 3] = vel[3:] * dt
        assert np.allclose(_mat(out), _mat(T) @ D, atol=1e-12)
    v9 = np.array([1.0, -2.0, 15.0, 0.1, -0.2, 0.7, 3.9, 1.5, 1.6])
    out, = kat(["minimal " + _fmt(v9)])
    assert np.allclose(_mat(out[:7])[:3, :3], Rotation.from_euler("ZYX", [v9[5], v9[4], v9[3]]).as_matrix(), atol=1e-12)   # yaw, pitch, roll
    assert np.allclose(out[:3], v9[:3]) and np.allclose(out[7:], v9[6:])


def test_unused_object_edge_reduces_to_the_object_ba_edge(kat):
    """EdgeStereoDynamicPointAndCuboid with Tcw = I == EdgeStereoSE3ProjectXYZ on (Tco, point): error and both Jacobians (the
    a16 kernels therefore cover it); with a real Tcw the point Jacobian is that of the combined pose."""
    rng = np.random.default_rng(22)
    bf = 384.38148
    cases, lines = [], []
    for _ in range(10):
        Tco = _p7(Rotation.from_euler("zyx", rng.uniform(-0.6, 0.6, 3)).as_matrix(), rng.uniform(-3, 3, 3) + [0, 0, 14])
        pt = rng.uniform(-2, 2, 3)
        obs = np.array([rng.uniform(0, 1242), rng.uniform(0, 375), rng.uniform(0, 1242)])
        cases.append((Tco, pt, obs))
        lines.append("edge " + _fmt(Tco, pt, obs, K4, bf, [0, 0, 0, 0, 0, 0, 1]))
    for (Tco, pt, obs), out in zip(cases, kat(lines)):
        err, Jp, Jx = oracle_lib.edge_eval(4, Tco, pt, obs, K4 + (bf,))        # E_STEREO_BA
        assert np.allclose(out[:3], err, rtol=0, atol=1e-9)
        assert np.allclose(out[3:21].reshape(3, 6), Jp, rtol=1e-12, atol=1e-9)
        assert np.allclose(out[21:30].reshape(3, 3), Jx, rtol=1e-12, atol=1e-9)
    # general Tcw: error and point Jacobian equal the a16 edge evaluated at the combined pose Tcw * Two
    Tcw = _p7(Rotation.from_euler("zyx", [0.1, -0.05, 0.02]).as_matrix(), np.array([0.3, -0.1, 0.5]))
    Two, pt, obs = cases[0]
    out, = kat(["edge " + _fmt(Two, pt, obs, K4, bf, Tcw)])
    M = _mat(Tcw) @ _mat(Two)
    comb = _p7(M[:3, :3], M[:3, 3])
    err, Jp, Jx = oracle_lib.edge_eval(4, comb, pt, obs, K4 + (bf,))
    assert np.allclose(out[:3], err, atol=1e-7) and np.allclose(out[21:30].reshape(3, 3), Jx, rtol=1e-9, atol=1e-7)
