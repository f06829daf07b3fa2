"""GPU parity of the ORB extractor (HIP path through the C-ABI) against the CPU oracle.
Bit-exact at every stage: pyramid planes, blurred planes, FAST candidates (order included),
quadtree selection (order included), angles (float bits), descriptors, final keypoints."""
import os

import numpy as np
import pytest

from oracle_lib import OracleORB, KEYPOINT_DTYPE

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _images():
    from PIL import Image
    from pointslot_amd import synth
    l, r = synth.stereo_pair()
    k = np.array(Image.open(os.path.join(GOLD, "kitti_000212_gray.png")))
    return {"synth_left": l, "synth_right": r, "kitti_000212": k}


@pytest.fixture(scope="module")
def images():
    return _images()


def _compare_stages(ex, orc, img, tag):
    h, w = img.shape
    for l in range(8):
        pg = ex.debug_read(0, l, 0, w, h)
        po = orc.padded(l)
        assert pg.shape == po.shape, (tag, l)
        assert np.array_equal(pg, po), "%s level %d pyramid plane differs at %d px" % (tag, l, (pg != po).sum())
        bg = ex.debug_read(0, l, 1, w, h)
        bo = orc.blur(l)
        assert np.array_equal(bg, bo), "%s level %d blur differs at %d px" % (tag, l, (bg != bo).sum())
        cg = ex.debug_read(0, l, 2, w, h)
        co = orc.candidates(l)
        assert len(cg) == len(co), "%s level %d: %d candidates vs oracle %d" % (tag, l, len(cg), len(co))
        assert np.array_equal(cg, co), "%s level %d candidate list differs" % (tag, l)
        sg = ex.debug_read(0, l, 3, w, h)
        ko = orc.level_keypoints(l)
        so = np.stack([ko["x"], ko["y"], ko["response"]], 1).astype(np.int32)
        assert len(sg) == len(so), "%s level %d: %d selected vs oracle %d" % (tag, l, len(sg), len(so))
        assert np.array_equal(sg, so), "%s level %d quadtree selection differs" % (tag, l)


@pytest.mark.parametrize("name", ["synth_left", "synth_right", "kitti_000212"])
@pytest.mark.parametrize("nfeatures", [2000, 1000])
def test_extract_bit_exact(images, name, nfeatures):
    from pointslot_amd.extractor import ORBextractor
    img = images[name]
    ex = ORBextractor(nfeatures, 1.2, 8, 20, 5)
    orc = OracleORB(nfeatures)
    kg, dg = ex(img)
    ko, do = orc.run(img)
    _compare_stages(ex, orc, img, name)
    assert len(kg) == len(ko)
    for f in KEYPOINT_DTYPE.names:
        assert np.array_equal(kg[f].view(np.uint32), ko[f].view(np.uint32)), "field %s differs" % f
    assert np.array_equal(dg, do)
    ex.close()


def test_getters_match_oracle():
    from pointslot_amd.extractor import ORBextractor
    ex = ORBextractor(2000, 1.2, 8, 20, 5)
    f, q, _ = OracleORB(2000).tables()
    assert np.array_equal(ex.GetScaleFactors(), f[0])
    assert np.array_equal(ex.GetInverseScaleFactors(), f[1])
    assert np.array_equal(ex.GetScaleSigmaSquares(), f[2])
    assert np.array_equal(ex.GetInverseScaleSigmaSquares(), f[3])
    assert np.array_equal(ex.features_per_level(), q)
    assert ex.GetLevels() == 8


def test_pyramid_contract_and_empty_image(images):
    from pointslot_amd.extractor import ORBextractor
    img = images["synth_left"]
    ex = ORBextractor(1000, 1.2, 8, 20, 5, keep_pyramid=True)
    kps, desc = ex(img)
    assert len(ex.mvImagePyramid) == 8
    assert np.array_equal(ex.mvImagePyramid[0], img)          # level 0 ROI is the input image
    orc = OracleORB(1000)
    orc.run(img)
    for l in range(8):
        assert np.array_equal(ex.mvImagePyramid[l], orc.padded(l)[19:-19, 19:-19])
    k0, d0 = ex(np.zeros((0, 0), np.uint8))                     # empty image -> silent return
    assert len(k0) == 0 and d0 is None
    flat = np.full((375, 1242), 77, np.uint8)                   # no corners -> descriptors released
    k1, d1 = ex(flat)
    assert len(k1) == 0 and d1 is None


def test_batch_device_matches_single(images):
    import torch
    from pointslot_amd.extractor import ORBextractor
    from pointslot_amd import synth
    batch = synth.stereo_batch(2)                                # 4 images
    d = torch.from_numpy(batch).cuda()
    ex = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=4)
    h, w = batch.shape[1:]
    ex.extract_batch_device(d.data_ptr(), 4, w, h, w, w * h)
    ex.sync()
    orc = OracleORB(2000)
    for i in range(4):
        kg, dg = ex.fetch(i)
        ko, do = orc.run(batch[i])
        assert len(kg) == len(ko)
        assert np.array_equal(kg.view(np.uint8), ko.view(np.uint8))
        assert np.array_equal(dg, do)
