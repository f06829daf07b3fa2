"""oracle/orb_oracle.cpp against the independently written numpy statement of the same pixel stages (tests/orb_second_opinion.py):
full 1242 x 375 synthetic image and the repository's real KITTI frame, every pyramid level - padded planes, blurred planes, FAST
candidate lists (order included) and the quadtree selection (order included).  CPU only."""
import os

import numpy as np
import pytest

import oracle_lib
import orb_second_opinion as so
from pointslot_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _images():
    from PIL import Image
    left, _ = synth.stereo_pair()
    kitti = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti_000212_gray.png")))
    return {"synthetic": left, "kitti": np.ascontiguousarray(kitti)}


@pytest.fixture(scope="module")
def runs():
    out = {}
    for name, img in _images().items():
        orc = oracle_lib.OracleORB(2000)
        orc.run(img)
        out[name] = (img, orc)
    return out


@pytest.mark.parametrize("name", ["synthetic", "kitti"])
def test_pyramid_border_and_blur(runs, name):
    img, orc = runs[name]
    levels = so.pyramid(img)
    for l, lv in enumerate(levels):
        assert orc.level_dims(l) == (lv.shape[1], lv.shape[0])
        assert np.array_equal(orc.padded(l), so.padded(lv)), "level %d: padded plane" % l
        assert np.array_equal(orc.blur(l), so.gaussian_blur7(lv)), "level %d: blurred plane" % l


@pytest.mark.parametrize("name", ["synthetic", "kitti"])
def test_fast_candidates_and_quadtree(runs, name):
    img, orc = runs[name]
    levels = so.pyramid(img)
    _, quota, _ = orc.tables()
    total = 0
    for l, lv in enumerate(levels):
        cand = so.fast_cells(lv)
        ref = orc.candidates(l)
        assert cand.shape == ref.shape, (l, cand.shape, ref.shape)
        assert np.array_equal(cand, ref), "level %d: FAST candidates (x, y, score) in emission order" % l
        h, w = lv.shape
        args = (so.EDGE - 3, w - so.EDGE + 3, so.EDGE - 3, h - so.EDGE + 3, int(quota[l]))
        sel = so.distribute(cand, *args)
        assert np.array_equal(sel, oracle_lib.distribute(ref, *args)), "level %d: quadtree selection" % l
        # ... and the extractor's own per-level result is that selection (pt += minBorder, ORBextractor.cc:842-849)
        kp = orc.level_keypoints(l)
        assert len(kp) == len(sel)
        assert np.array_equal(np.stack([kp["x"], kp["y"], kp["response"]], 1).astype(np.int32), sel + np.array([so.EDGE - 3, so.EDGE - 3, 0], np.int32))
        total += len(sel)
    assert total >= 1900


def test_masked_object_features_definition():
    """the 8f-2 stand-in (oracle: orc_orb_run_masked): FAST candidates whose level-0 pixel lies outside the mask are dropped before
    the quadtree - restated here on the second opinion's candidates"""
    img = _images()["kitti"]
    mask = np.zeros_like(img)
    mask[80:330, 300:900] = 255
    mask[100:200, 500:700] = 0                                   # a hole
    orc = oracle_lib.OracleORB(1000)
    kps, _ = orc.run_masked(img, mask)
    sf, _ = so.scale_tables()
    _, quota, _ = orc.tables()
    total = 0
    for l, lv in enumerate(so.pyramid(img)):
        cand = so.fast_cells(lv)
        lx = np.rint((cand[:, 0] + so.EDGE - 3).astype(np.float32) * sf[l]).astype(int).clip(0, img.shape[1] - 1)
        ly = np.rint((cand[:, 1] + so.EDGE - 3).astype(np.float32) * sf[l]).astype(int).clip(0, img.shape[0] - 1)
        cand = cand[mask[ly, lx] != 0]
        h, w = lv.shape
        sel = so.distribute(cand, so.EDGE - 3, w - so.EDGE + 3, so.EDGE - 3, h - so.EDGE + 3, int(quota[l])) if len(cand) else np.zeros((0, 3), np.int32)
        kp = orc.level_keypoints(l)
        assert len(kp) == len(sel), (l, len(kp), len(sel))
        if len(sel):
            assert np.array_equal(np.stack([kp["x"], kp["y"], kp["response"]], 1).astype(np.int32), sel + np.array([so.EDGE - 3, so.EDGE - 3, 0], np.int32))
        total += len(sel)
    assert total == len(kps) and total > 300
