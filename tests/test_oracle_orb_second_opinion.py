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


def test_cv_orb_stages_second_opinion():
    """the restatement of OpenCV's own ORB (object features): INTER_LINEAR_EXACT pyramid, whole-image FAST with the
    mask / border filters, and the Harris responses against the numpy statement of the same definitions"""
    img = _images()["kitti"]
    mask = np.zeros_like(img)
    mask[60:330, 250:1000] = 255
    orc = oracle_lib.OracleCvORB()
    kps, desc = orc.run(img, mask)
    assert 900 <= len(kps) <= 1100
    levels = so.cv_pyramid(img)
    mlevels = [mask]
    for l in range(1, 8):
        m = so.resize_linear_exact_u8(mlevels[-1], levels[l].shape[1], levels[l].shape[0])
        mlevels.append(np.where(m > 254, m, 0).astype(np.uint8))
    for l, lv in enumerate(levels):
        assert orc.level_dims(l) == (lv.shape[1], lv.shape[0])
        assert np.array_equal(orc.plane(l, 0), lv), "level %d image" % l
        assert np.array_equal(orc.plane(l, 1), so.gaussian_blur7(lv)), "level %d blurred" % l
        assert np.array_equal(orc.plane(l, 2), mlevels[l]), "level %d mask" % l
        f = orc.fast(l)
        mine = so.cv_fast(lv, 20, 19, mlevels[l])
        assert np.array_equal(f[:, :3].astype(np.int32), mine), "level %d FAST keypoints" % l
        hr = so.cv_harris(lv, mine[:, 0], mine[:, 1])
        assert np.array_equal(f[:, 3].view(np.uint32), hr.view(np.uint32)), "level %d Harris responses" % l
    # per level: the quota's best by Harris response among the 2 x quota best by FAST score (as sets: the order is std::nth_element's)
    sf = np.array([np.float32(np.float64(np.float32(1.2)) ** l) for l in range(8)], np.float32)
    for l in range(8):
        sel = kps[kps["octave"] == l]
        if len(sel) == 0:
            continue
        f = orc.fast(l)
        lx = np.rint(sel["x"] / sf[l]).astype(int); ly = np.rint(sel["y"] / sf[l]).astype(int)
        table = {(int(x), int(y)): (s, hsc) for x, y, s, hsc in f}
        assert all((x, y) in table for x, y in zip(lx, ly))
        assert np.array_equal(np.array([table[(x, y)][1] for x, y in zip(lx, ly)], np.float32).view(np.uint32), sel["response"].view(np.uint32))


# ---- r04: orientation, descriptor, stereo matcher, cv::RNG --------------------------------------------------------
def _pattern():
    import re
    txt = open(os.path.join(ROOT, "oracle", "orb_pattern.inc")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    v = np.array([int(t) for t in re.findall(r"-?\d+", txt)], np.int64)
    assert v.size == 1024
    return v


@pytest.mark.parametrize("name", ["synthetic", "kitti"])
def test_orientation_bits(runs, name):
    """IC_Angle's integer moments over the 31-pixel disc and OpenCV's fastAtan2 polynomial, float bits of every keypoint's angle"""
    img, orc = runs[name]
    n = 0
    for l in range(8):
        kp = orc.level_keypoints(l)
        if len(kp) == 0:
            continue
        ang = so.ic_angles(orc.padded(l), kp["x"], kp["y"])
        assert np.array_equal(ang.view(np.uint32), kp["angle"].view(np.uint32)), "level %d: %d of %d angles differ" % (
            l, int((ang.view(np.uint32) != kp["angle"].view(np.uint32)).sum()), len(kp))
        n += len(kp)
    assert n >= 1900
    # the polynomial against the mathematical atan2 (its documented accuracy) on a sweep that includes the octant boundaries
    t = np.linspace(-np.pi, np.pi, 7201)
    got = so.fast_atan2_deg(np.sin(t).astype(np.float32) * 100, np.cos(t).astype(np.float32) * 100)
    want = np.degrees(np.arctan2(np.sin(t), np.cos(t))) % 360.0
    d = np.abs(((got - want) + 180.0) % 360.0 - 180.0)
    assert d.max() < 0.3


@pytest.mark.parametrize("name", ["synthetic", "kitti"])
def test_steered_brief_bits(runs, name):
    """the 256 steered comparisons of every keypoint (float32 rotation, round-half-to-even taps) on the blurred level"""
    img, orc = runs[name]
    kps, desc = orc.run(img)
    pat = _pattern()
    o = 0
    for l in range(8):
        kp = orc.level_keypoints(l)
        if len(kp) == 0:
            continue
        mine = so.steered_brief(orc.blur(l), kp["x"], kp["y"], kp["angle"], pat)
        assert np.array_equal(mine, desc[o:o + len(kp)]), "level %d: %d of %d descriptors differ" % (l, int((mine != desc[o:o + len(kp)]).any(1).sum()), len(kp))
        assert np.array_equal(kps["octave"][o:o + len(kp)], np.full(len(kp), l))
        o += len(kp)
    assert o == len(kps)
    # the rounding rule itself: cvRound is round-half-to-even (a tap at exactly .5 goes to the even pixel)
    assert [int(np.rint(np.float32(v))) for v in (0.5, 1.5, 2.5, -0.5, -1.5)] == [0, 2, 2, 0, -2]


def test_stereo_matcher_second_opinion():
    """Frame::ComputeStereoMatches restated from the reference's text in numpy against the oracle's: uRight / depth float bits, the
    kept count - on the synthetic pair and on the KITTI frame against a shifted copy of itself"""
    from PIL import Image
    left, right = synth.stereo_pair()
    kitti = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti_000212_gray.png"))))
    shifted = np.roll(kitti, -9, axis=1).copy()
    shifted[:, -9:] = kitti[:, -1:]
    for tag, (a, b) in {"synthetic": (left, right), "kitti": (kitti, shifted)}.items():
        ol, orr = oracle_lib.OracleORB(2000), oracle_lib.OracleORB(2000)
        kl, dl = ol.run(a)
        kr, dr = orr.run(b)
        mb, mbf = np.float32(0.5327), np.float32(384.38148)
        kept, ur, dp = oracle_lib.stereo_match(ol, orr, float(mb), float(mbf))
        sf, isf = so.scale_tables()
        pl = [ol.padded(l)[so.EDGE:-so.EDGE, so.EDGE:-so.EDGE] for l in range(8)]
        pr = [orr.padded(l)[so.EDGE:-so.EDGE, so.EDGE:-so.EDGE] for l in range(8)]
        k2, ur2, dp2 = so.stereo_matches(kl, dl, kr, dr, pl, pr, sf, isf, mb, mbf)
        assert kept == k2 and kept > 300, (tag, kept, k2)
        assert np.array_equal(ur.view(np.uint32), ur2.view(np.uint32)), (tag, int((ur.view(np.uint32) != ur2.view(np.uint32)).sum()))
        assert np.array_equal(dp.view(np.uint32), dp2.view(np.uint32)), (tag, int((dp.view(np.uint32) != dp2.view(np.uint32)).sum()))


def test_cv_rng_second_opinion():
    """cv::RNG's multiply-with-carry stream (the RANSAC centroid of TrackMapObject draws from it, Tracking.cc:1664,1817) against the
    closed form of the same generator: s_k = s_0 2^(-32 k) mod (a 2^32 - 1) - no update rule shared with the restatement"""
    from pointslot_amd.object_tracker import CvRng
    ref = so.cv_rng_closed_form(5000)
    r = CvRng()
    for k in range(5000):
        n = 1 + (k * 7919) % 1000
        assert r(n) == int(ref[k]) % n, k
    # the generator's invariants: the modulus is what makes it a full-period Lehmer sequence in disguise
    a, b = 4164903690, 1 << 32
    assert pow(b, -1, a * b - 1) * b % (a * b - 1) == 1
