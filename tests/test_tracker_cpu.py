"""End-to-end data flow of the hot path on a generated mini-sequence, CPU checker behind the
loop: the chain extraction -> stereo -> projection matching -> pose optimisation recovers the generating trajectory."""
import os

import numpy as np
import pytest

from pointslot_amd import sequence
from pointslot_amd.tracker import StereoOdometry, save_trajectory_kitti
from oracle_backend import OracleBackend


@pytest.fixture(scope="module")
def seq():
    return sequence.generate(n_frames=6, seed=4, w=800, h=300)


def run(seq, backend, **kw):
    h, w = seq["left"][0].shape
    vo = StereoOdometry(backend, seq["K"], seq["bf"], w, h, **kw)
    for l, r in zip(seq["left"], seq["right"]):
        vo.track(l, r)
    return vo


def test_sequence_layout_round_trip(tmp_path, seq):
    d = str(tmp_path / "0000")
    sequence.write(d, seq)
    for sub in ("image_02/000005.png", "image_03/000000.png", "Segmentation/000003.png", "timestamp.txt", "ObjectTracking.txt", "poses.txt"):
        assert os.path.exists(os.path.join(d, sub)), sub
    back = sequence.load(d)
    assert len(back["left"]) == 6
    assert np.array_equal(back["left"][2], seq["left"][2]) and np.array_equal(back["right"][5], seq["right"][5])
    rows = [l.split() for l in open(os.path.join(d, "ObjectTracking.txt"))]
    assert len(rows) == 12 and all(len(r) == 17 for r in rows)                # 2 boxes x 6 frames, KITTI tracking columns
    assert abs(back["calib"]["Camera.bf"] - seq["bf"]) < 1e-4


def test_gray_conversion_matches_fixed_point_formula():
    a = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    g = sequence._to_gray(a, rgb_order=True)[0]
    assert list(g) == [(255 * 4899 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 1868 + 8192) >> 14, (10 * 4899 + 20 * 9617 + 30 * 1868 + 8192) >> 14]
    assert sequence._to_gray(a, rgb_order=False)[0][0] == (255 * 1868 + 8192) >> 14


@pytest.mark.parametrize("local_map", [False, True])
def test_odometry_recovers_ground_truth(seq, tmp_path, local_map):
    vo = run(seq, OracleBackend(), track_local_map=local_map)
    assert vo.state == "OK" and all(t is not None for t in vo.trajectory)
    for k, tcw in enumerate(vo.trajectory):
        twc = -(tcw[:3, :3].T @ tcw[:3, 3])
        gt = seq["twc"][k][:, 3]
        assert np.abs(twc - gt).max() < 0.03, (k, twc, gt)                     # 3 cm over a 0.4 m path, sub-pixel stereo noise
        assert np.abs(tcw[:3, :3] - np.eye(3)).max() < 2e-3
    assert all(s["matches"] > 100 for s in vo.stats[1:])
    p = str(tmp_path / "CameraTrajectory.txt")
    save_trajectory_kitti(p, vo.trajectory)
    rows = np.loadtxt(p)
    assert rows.shape == (6, 12)
    assert np.allclose(rows[:, 3], seq["twc"][:, 0, 3], atol=0.03)


def test_too_few_features_does_not_initialise():
    flat = np.full((300, 800), 90, np.uint8)
    vo = StereoOdometry(OracleBackend(), (721.5377, 721.5377, 609.5593, 172.854), 384.38148, 800, 300)
    assert vo.track(flat, flat) is None and vo.state == "NOT_INITIALIZED"
