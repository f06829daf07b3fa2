"""The object-feature detector: cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, kp, descriptor)
(/root/reference/src/Frame.cc:2623-2627) on the GPU (ps_cvorb_*) against the CPU restatement of OpenCV 3.4.3's ORB_Impl
(oracle/orb_oracle.cpp: cv_orb_run) - every level's image, mask and blurred plane, the FAST keypoints after the mask / border
filters with their Harris responses, and the final keypoints (order included) and descriptors, bit for bit."""
import numpy as np
import pytest

from oracle_lib import OracleCvORB
from pointslot_amd import sequence, synth

pytestmark = pytest.mark.gpu


def _compare(img, mask, **kw):
    from pointslot_amd.object_orb import ORB
    det = ORB(**kw)
    orc = OracleCvORB(kw.get("nfeatures", 1000), kw.get("scaleFactor", 1.2), kw.get("nlevels", 8), kw.get("edgeThreshold", 19), kw.get("fastThreshold", 20))
    kps, desc = det.detectAndCompute(img, mask)
    ko, do = orc.run(img, mask)
    for l in range(kw.get("nlevels", 8)):
        assert det.level_size(l) == orc.level_dims(l)
        assert np.array_equal(det.debug_plane(l, 0), orc.plane(l, 0)), "level %d image (INTER_LINEAR_EXACT)" % l
        assert np.array_equal(det.debug_plane(l, 1), orc.plane(l, 1)), "level %d blurred" % l
        if mask is not None:
            assert np.array_equal(det.debug_plane(l, 2), orc.plane(l, 2)), "level %d mask" % l
        fg, fo = det.debug_fast(l), orc.fast(l)
        assert fg.shape == fo.shape, (l, fg.shape, fo.shape)
        assert np.array_equal(fg.view(np.uint32), fo.view(np.uint32)), "level %d FAST keypoints / Harris responses" % l
    assert len(kps) == len(ko), (len(kps), len(ko))
    assert np.array_equal(kps.view(np.uint8), ko.view(np.uint8)), "keypoints (order included)"
    assert np.array_equal(desc, do)
    det.close()
    return kps


def test_object_orb_with_instance_mask():
    seq = sequence.generate(n_frames=1, seed=9)
    img = seq["left"][0]
    mask = np.where(seq["seg"][0] != 0, 255, 0).astype(np.uint8)
    kps = _compare(img, mask)
    assert len(kps) > 30
    inside = mask[np.clip(np.rint(kps["y"]).astype(int), 0, img.shape[0] - 1), np.clip(np.rint(kps["x"]).astype(int), 0, img.shape[1] - 1)]
    assert (inside != 0).mean() > 0.95          # (a level-7 keypoint scaled back may land a pixel outside)


def test_object_orb_whole_image_and_real_frame():
    import os
    from PIL import Image
    left, _ = synth.stereo_pair()
    kps = _compare(left, None)
    assert 900 <= len(kps) <= 1100 and set(np.unique(kps["octave"])) == set(range(8))
    kitti = np.ascontiguousarray(np.asarray(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_000212_gray.png"))))
    big = np.zeros_like(kitti); big[100:340, 200:1100] = 255
    kps = _compare(kitti, big)
    assert len(kps) >= 900


def test_object_orb_other_parameters_and_empty_mask():
    left, _ = synth.stereo_pair(w=640, h=360)
    _compare(left, None, nfeatures=300, scaleFactor=1.3, nlevels=5, edgeThreshold=12, fastThreshold=12)
    from pointslot_amd.object_orb import ORB
    det = ORB()
    kps, desc = det.detectAndCompute(left, np.zeros_like(left))
    assert len(kps) == 0
    det.close()
