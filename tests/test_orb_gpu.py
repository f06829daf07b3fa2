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


@pytest.mark.parametrize("kind", ["uniform", "salt", "mixed"])
def test_noise_images_take_the_multi_round_path(kind):
    """white noise lists most pixels of a FAST cell as dark AND bright survivors: the survivor region of orb_fast_cells
    overflows and the cell is scored in several rounds with the row-wise NMS - every stage stays bit-exact"""
    from pointslot_amd.extractor import ORBextractor
    rng = np.random.default_rng(11)
    h, w = 375, 1242
    if kind == "uniform":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "salt":
        img = np.where(rng.random((h, w)) < 0.5, 20, 235).astype(np.uint8)
    else:   # left half noise, right half flat with a few corners: cells of both kinds in one launch
        img = np.full((h, w), 128, np.uint8)
        img[:, : w // 2] = rng.integers(0, 256, (h, w // 2), dtype=np.uint8)
        img[50:200, 800:1000] = 30
    ex = ORBextractor(2000, 1.2, 8, 20, 5)
    orc = OracleORB(2000)
    kg, dg = ex(img)
    ko, do = orc.run(img)
    _compare_stages(ex, orc, img, kind)
    assert len(kg) == len(ko) and len(kg) > 100
    assert np.array_equal(kg.view(np.uint8), ko.view(np.uint8)) and np.array_equal(dg, do)
    ex.close()


def test_noise_image_with_wide_cells():
    """the same overflow path in the kernel instance for FAST cells wider than 32 pixels (16 four-pixel groups per row)"""
    from pointslot_amd.extractor import ORBextractor
    rng = np.random.default_rng(5)
    h, w = 200, 320                                      # level 1 has 34-pixel cells
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[:, 200:] = (img[:, 200:] // 8 + 100).astype(np.uint8)   # a low-contrast part: cells that take the minThFAST pass
    ex = ORBextractor(600, 1.2, 4, 20, 7)
    orc = OracleORB(600, 1.2, 4, 20, 7)
    kg, dg = ex(img)
    ko, do = orc.run(img)
    assert len(kg) == len(ko) and len(kg) > 50
    assert np.array_equal(kg.view(np.uint8), ko.view(np.uint8)) and np.array_equal(dg, do)
    ex.close()


@pytest.mark.parametrize("shape,nfeatures,nlevels,ths,scale", [
    ((480, 640), 1500, 8, (20, 7), 1.2),
    ((300, 800), 500, 8, (20, 5), 1.2),
    ((376, 1241), 3000, 8, (12, 5), 1.2),
    ((375, 1242), 1200, 5, (20, 5), 1.2),
    ((200, 320), 300, 4, (30, 10), 1.2),
    ((479, 641), 1000, 8, (20, 7), 1.2),        # odd sizes: partial dword groups and tiles on every edge
    ((1080, 1920), 2000, 8, (20, 7), 1.2),      # many tiles per level
    ((600, 800), 800, 4, (20, 7), 1.5),         # other scale factors: different tap spans per lane
    ((700, 900), 600, 3, (20, 7), 2.5),         # span > 8 source bytes: the generic gather path of the level kernel
    ((480, 752), 1000, 8, (20, 7), 1.1),
])
def test_other_shapes_and_parameters(shape, nfeatures, nlevels, ths, scale):
    """parity does not depend on the KITTI geometry: other sizes, level counts, quotas, FAST thresholds and scale factors"""
    from pointslot_amd import synth
    from pointslot_amd.extractor import ORBextractor
    h, w = shape
    img, _ = synth.stereo_pair(seed=0x77 + h, w=w, h=h)
    ex = ORBextractor(nfeatures, scale, nlevels, ths[0], ths[1])
    orc = OracleORB(nfeatures, scale, nlevels, ths[0], ths[1])
    kg, dg = ex(img)
    ko, do = orc.run(img)
    assert len(kg) == len(ko) and len(kg) > 50
    assert np.array_equal(kg.view(np.uint8), ko.view(np.uint8))
    assert np.array_equal(dg, do)
    ex.close()


def test_replan_on_size_change_strided_input_and_determinism(images):
    from pointslot_amd.extractor import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 5)
    big = images["synth_left"]
    k1, d1 = ex(big)
    small = np.ascontiguousarray(big[:300, :900])
    k2, d2 = ex(small)                                         # new geometry on the same handle
    view = big[:300, :900]                                     # same pixels, row stride 1242
    k3, d3 = ex(view)
    assert np.array_equal(k2.view(np.uint8), k3.view(np.uint8)) and np.array_equal(d2, d3)
    k4, d4 = ex(big)                                           # back to the first geometry: identical to the first run
    assert np.array_equal(k1.view(np.uint8), k4.view(np.uint8)) and np.array_equal(d1, d4)
    ko, do = OracleORB(1000).run(small)
    assert np.array_equal(k2.view(np.uint8), ko.view(np.uint8)) and np.array_equal(d2, do)
    # structural properties at full size: level-major order, octave range, response = FAST score
    assert np.all(np.diff(k1["octave"]) >= 0) and k1["octave"].min() == 0 and k1["octave"].max() == 7
    assert np.all(k1["response"] >= 5) and np.all(k1["class_id"] == -1)
    ex.close()


def test_two_extractors_on_two_threads(images):
    """Frame::Frame runs the left and the right extractor on two std::threads (Frame.cc:709-710): handles are independent"""
    import threading
    from pointslot_amd.extractor import ORBextractor
    exs = [ORBextractor(2000, 1.2, 8, 20, 5), ORBextractor(2000, 1.2, 8, 20, 5)]
    imgs = [images["synth_left"], images["synth_right"]]
    out = [None, None]

    def work(i):
        for _ in range(3):
            out[i] = exs[i](imgs[i])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(2):
        ko, do = OracleORB(2000).run(imgs[i])
        assert np.array_equal(out[i][0].view(np.uint8), ko.view(np.uint8)) and np.array_equal(out[i][1], do)
        exs[i].close()


def test_full_size_batch_properties():
    """BASELINE configs[1] at the bench's batch size: every image of a 64-pair batch matches its single-image run
    (checksum of checksums), and descriptors are self-consistent (Hamming(a, a) = 0 on the diagonal)."""
    import hashlib
    import torch
    from pointslot_amd import synth
    from pointslot_amd.extractor import ORBextractor
    from pointslot_amd.matcher import ORBmatcher
    batch = synth.stereo_batch(8)                              # 16 images through the batch path
    d = torch.from_numpy(batch).cuda()
    exb = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=16)
    ex1 = ORBextractor(2000, 1.2, 8, 20, 5)
    hgt, w = batch.shape[1:]
    exb.extract_batch_device(d.data_ptr(), 16, w, hgt, w, w * hgt)
    hb, h1 = hashlib.sha256(), hashlib.sha256()
    for i in range(16):
        kb, db = exb.fetch(i)
        k1, d1 = ex1(batch[i])
        hb.update(kb.tobytes()); hb.update(db.tobytes())
        h1.update(k1.tobytes()); h1.update(d1.tobytes())
    assert hb.hexdigest() == h1.hexdigest()
    m = ORBmatcher(0.9, True)
    D = m.DescriptorDistanceMatrix(db[:500], db[:500])
    assert np.all(np.diag(D) == 0) and np.array_equal(D, D.T)
    m.close(); exb.close(); ex1.close()


def test_large_batch_takes_the_split_quadtree_launch(images):
    """from 256 images on the quadtree runs as two launches (512-thread workgroups for the large levels, 256-thread ones with half
    the LDS for the small levels): images of such a batch equal their single-image runs, real texture and synthetic"""
    import torch
    from pointslot_amd.extractor import ORBextractor
    k = images["kitti_000212"]
    hgt, w = images["synth_left"].shape
    kk = np.zeros((hgt, w), np.uint8)
    kk[: min(hgt, k.shape[0]), : min(w, k.shape[1])] = k[:hgt, :w]
    distinct = [images["synth_left"], kk, images["synth_right"], np.ascontiguousarray(kk[:, ::-1])]
    n = 256
    batch = np.stack([distinct[i % 4] for i in range(n)])
    d = torch.from_numpy(batch).cuda()
    exb = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=n)
    ex1 = ORBextractor(2000, 1.2, 8, 20, 5)
    exb.extract_batch_device(d.data_ptr(), n, w, hgt, w, w * hgt)
    ref = [ex1(im) for im in distinct]
    for i in (0, 1, 2, 3, 129, 254, 255):
        kb, db = exb.fetch(i)
        k1, d1 = ref[i % 4]
        assert len(kb) == len(k1) and np.array_equal(kb.view(np.uint8), k1.view(np.uint8)) and np.array_equal(db, d1), "image %d of the batch differs" % i
    exb.close(); ex1.close()


def test_stereo_matches_bit_exact(images):
    """Frame::ComputeStereoMatches: mvuRight / mvDepth from the device-resident pyramids and descriptors are
    bit-exact against the oracle, through both layouts (two extractor objects; one interleaved batch)."""
    import torch
    from oracle_lib import stereo_match
    from pointslot_amd import synth
    from pointslot_amd.extractor import ORBextractor, ComputeStereoMatches
    bf, fx = 384.38148, 721.5377
    mb, mbf = np.float32(bf / fx), np.float32(bf)
    L, R = images["synth_left"], images["synth_right"]
    exl, exr = ORBextractor(2000, 1.2, 8, 20, 5), ORBextractor(2000, 1.2, 8, 20, 5)
    kl, _ = exl(L); exr(R)
    ur, dp = ComputeStereoMatches(exl, exr, mb, mbf)
    ol, orr = OracleORB(2000), OracleORB(2000)
    ol.run(L); orr.run(R)
    kept, uro, dpo = stereo_match(ol, orr, mb, mbf)
    assert len(ur) == len(uro) == len(kl) and kept > 500
    assert np.array_equal(ur.view(np.uint32), uro.view(np.uint32)), int((ur != uro).sum())
    assert np.array_equal(dp.view(np.uint32), dpo.view(np.uint32))
    assert (ur >= 0).sum() == kept
    # batch layout: two pairs
    batch = synth.stereo_batch(2)
    d = torch.from_numpy(batch).cuda()
    exb = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=4)
    h, w = batch.shape[1:]
    exb.extract_batch_device(d.data_ptr(), 4, w, h, w, w * h)
    exb.stereo_match_batch(2, mb, mbf)
    for k in range(2):
        ol.run(batch[2 * k]); orr.run(batch[2 * k + 1])
        kept, uro, dpo = stereo_match(ol, orr, mb, mbf)
        urb, dpb, keptb = exb.stereo_fetch(k)
        assert keptb == kept
        assert np.array_equal(urb.view(np.uint32), uro.view(np.uint32)) and np.array_equal(dpb.view(np.uint32), dpo.view(np.uint32))
    # depth is physically sensible: the synthetic right image is the left warped by d(y) = bf / z(y)
    m = ur >= 0
    ztrue = 60 + (6 - 60) * (kl["y"][m] / 374.0)
    assert np.median(np.abs(dp[m] - ztrue) / ztrue) < 0.1
    exl.close(); exr.close(); exb.close()


def test_host_image_batch_and_bulk_frame_fetch(images):
    """ps_orb_extract_batch (host images, pinned or pageable) + ps_orb_stereo_match_batch + ps_orb_stereo_fetch_frames: what a
    batch of stereo Frames keeps (mvKeys, mDescriptors, mvuRight, mvDepth) equals the oracle's and the per-image entry points',
    including a pair of blank images (no keypoints) in the middle of the batch and a second batch on the same handle."""
    from oracle_lib import stereo_match
    from pointslot_amd import synth
    from pointslot_amd._lib import PinnedBuffer
    from pointslot_amd.extractor import ORBextractor
    bf, fx = 384.38148, 721.5377
    mb, mbf = np.float32(bf / fx), np.float32(bf)
    batch = synth.stereo_batch(2)
    h, w = batch.shape[1:]
    blank = np.full((h, w), 90, np.uint8)
    imgs = [batch[0], batch[1], blank, blank, batch[2], batch[3]]
    pin = PinnedBuffer(len(imgs) * h * w)
    pinned = pin.array.reshape(len(imgs), h, w)
    pinned[:] = np.stack(imgs)
    ex = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=6)
    ol, orr = OracleORB(2000), OracleORB(2000)
    for source in ([pinned[i] for i in range(6)], imgs):          # page-locked, then pageable host memory
        ex.extract_batch(source)
        ex.stereo_match_batch(3, mb, mbf)
        frames = ex.stereo_fetch_frames(3)
        for k, (kps, desc, ur, dp, kept) in enumerate(frames):
            ko, do = ol.run(imgs[2 * k]); orr.run(imgs[2 * k + 1])
            if k == 1:
                assert len(kps) == 0 and len(ko) == 0 and kept == 0
                continue
            kept_o, uro, dpo = stereo_match(ol, orr, mb, mbf)
            assert len(kps) == len(ko) > 1000 and kept == kept_o
            assert np.array_equal(kps.view(np.uint8), ko.view(np.uint8)) and np.array_equal(desc, do)
            assert np.array_equal(ur.view(np.uint32), uro.view(np.uint32)) and np.array_equal(dp.view(np.uint32), dpo.view(np.uint32))
            k1, d1 = ex.fetch(2 * k)
            u1, p1, kept1 = ex.stereo_fetch(k)
            assert np.array_equal(k1.view(np.uint8), kps.view(np.uint8)) and np.array_equal(d1, desc)
            assert np.array_equal(u1.view(np.uint32), ur.view(np.uint32)) and np.array_equal(p1.view(np.uint32), dp.view(np.uint32)) and kept1 == kept
    # a new extraction invalidates the previous stereo results until the matcher has run again
    ex.extract_batch(imgs[:2])
    with pytest.raises(Exception):
        ex.stereo_fetch_frames(1)
    ex.close(); pin.close()


def test_object_stereo_matches_bit_exact(images):
    """Frame::ComputeObjStereoMatches (Frame.cc:2318-2503): caller-provided object key sets (here: the frame's keypoints inside
    a detection box, in shuffled order) matched against the device-resident pyramids; bit-exact against the checker."""
    from oracle_lib import stereo_match_keys
    from pointslot_amd.extractor import ORBextractor, ComputeObjStereoMatches, ComputeStereoMatches
    bf, fx = 384.38148, 721.5377
    mb, mbf = np.float32(bf / fx), np.float32(bf)
    L, R = images["synth_left"], images["synth_right"]
    exl, exr = ORBextractor(2000, 1.2, 8, 20, 5), ORBextractor(2000, 1.2, 8, 20, 5)
    kl, dl = exl(L); kr, dr = exr(R)
    ol, orr = OracleORB(2000), OracleORB(2000)
    ol.run(L); orr.run(R)
    ur_frame, dp_frame = ComputeStereoMatches(exl, exr, mb, mbf)
    rng = np.random.default_rng(5)
    for box in [(300, 60, 900, 330), (0, 0, 1242, 375), (500, 150, 560, 200), (2000, 0, 2100, 10)]:
        sl = np.nonzero((kl["x"] >= box[0]) & (kl["x"] < box[2]) & (kl["y"] >= box[1]) & (kl["y"] < box[3]))[0]
        sr = np.nonzero((kr["x"] >= box[0] - 80) & (kr["x"] < box[2]) & (kr["y"] >= box[1] - 3) & (kr["y"] < box[3] + 3))[0]
        sl, sr = rng.permutation(sl), rng.permutation(sr)
        ur, dp, kept = ComputeObjStereoMatches(exl, exr, kl[sl], dl[sl], kr[sr], dr[sr], mb, mbf)
        ko, uro, dpo = stereo_match_keys(ol, orr, kl[sl], dl[sl], kr[sr], dr[sr], mb, mbf)
        assert kept == ko and len(ur) == len(sl)
        assert np.array_equal(ur.view(np.uint32), uro.view(np.uint32)) and np.array_equal(dp.view(np.uint32), dpo.view(np.uint32))
        if box == (300, 60, 900, 330):
            assert kept > 100
    # the object call leaves the frame's own results fetchable and unchanged
    ur2, dp2 = ComputeStereoMatches(exl, exr, mb, mbf)
    assert np.array_equal(ur2.view(np.uint32), ur_frame.view(np.uint32))
    exl.close(); exr.close()


def test_randomised_extractor_sweep():
    """random image sizes (tile and dword-group remainders on every edge), strided views, level counts, scale factors, quotas,
    thresholds and low-texture images where most cells take the minThFAST pass: keypoints and descriptors stay bit-exact"""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_orb.py")
    r = subprocess.run([sys.executable, tool, "5", "14"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_object_features_masked_stand_in():
    """SURVEY.md 8f-2, the declared stand-in for cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, ...)
    (Frame.cc:2623-2665): this extractor with the keypoints outside the instance mask dropped before the quadtree - bit-exact
    against the CPU restatement of the same definition; a full mask equals the plain extractor, an empty one yields nothing."""
    from pointslot_amd import sequence
    from pointslot_amd.extractor import ORBextractor
    seq = sequence.generate(n_frames=1, seed=9)
    img = seq["left"][0]
    mask = np.where((seq["seg"][0] != 0), 255, 0).astype(np.uint8)          # the two boxes' instance pixels
    assert 2000 < (mask != 0).sum() < 40000
    ex = ORBextractor(1000, 1.2, 8, 20, 5)
    orc = OracleORB(1000)
    big = np.zeros_like(mask); big[60:300, 200:1000] = 7                      # any non-zero value counts
    for m in (mask, big):
        kps, desc = ex.detect_masked(img, m)
        ko, do = orc.run_masked(img, m)
        assert len(kps) == len(ko) and len(kps) > 20
        assert np.array_equal(kps.view(np.uint8), ko.view(np.uint8)) and np.array_equal(desc, do)
        lx = np.clip(np.rint(kps["x"]).astype(int), 0, img.shape[1] - 1); ly = np.clip(np.rint(kps["y"]).astype(int), 0, img.shape[0] - 1)
        assert np.all(m[ly, lx] != 0)                                          # every keypoint lies inside the mask
    # the quota is spent inside the mask: more in-mask keypoints than the plain extractor leaves there
    plain, _ = ex(img)
    px = np.clip(np.rint(plain["x"]).astype(int), 0, img.shape[1] - 1); py = np.clip(np.rint(plain["y"]).astype(int), 0, img.shape[0] - 1)
    kb, _ = ex.detect_masked(img, big)
    assert len(kb) >= (big[py, px] != 0).sum()
    full, dfull = ex.detect_masked(img, np.full_like(mask, 255))
    pk, pd = ex(img)
    assert np.array_equal(full.view(np.uint8), pk.view(np.uint8)) and np.array_equal(dfull, pd)
    none, _ = ex.detect_masked(img, np.zeros_like(mask))
    assert len(none) == 0
    ex.close()
