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


def _blob_mask(rng, h, w, nblobs, big=False):
    m = np.zeros((h, w), np.uint8)
    for _ in range(nblobs):
        bw, bh = (int(rng.integers(150, 500)), int(rng.integers(80, 220))) if big else (int(rng.integers(20, 160)), int(rng.integers(15, 90)))
        x0, y0 = int(rng.integers(-bw // 2, w - bw // 2)), int(rng.integers(-bh // 2, h - bh // 2))
        yy, xx = np.mgrid[0:h, 0:w]
        if rng.random() < 0.5:
            m[max(y0, 0):y0 + bh, max(x0, 0):x0 + bw] = 255
        else:
            m[((xx - x0 - bw / 2) / (bw / 2)) ** 2 + ((yy - y0 - bh / 2) / (bh / 2)) ** 2 <= 1] = 255
    return m


def test_batched_device_detector_matches_the_cpu_restatement():
    """ps_cvorb_detect_batch_device: images + object masks in HBM, tile-restricted work, retainBest on the device - keypoints (order
    included) and descriptors of every image against cv_orb_run; masks: boxes / ellipses of all sizes, touching the image borders,
    an empty one, a full one small enough for the per-level limit, and large ones that push levels over their quota (both
    retainBest steps run, ties included)."""
    import os
    import torch
    from PIL import Image
    from pointslot_amd.object_orb import ORB
    kitti = np.ascontiguousarray(np.asarray(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_000212_gray.png"))))
    h, w = kitti.shape
    rng = np.random.default_rng(5)
    seq = sequence.generate(n_frames=3, seed=11, w=w, h=h, texture=sequence.kitti_texture())
    imgs, masks = [], []
    for k in range(3):
        imgs.append(seq["left"][k]); masks.append(np.where(seq["seg"][k] != 0, 255, 0).astype(np.uint8))
    for k in range(9):
        imgs.append(kitti if k % 2 == 0 else np.ascontiguousarray(kitti[:, ::-1]))
        masks.append(_blob_mask(rng, h, w, 1 + k % 4, big=k >= 5))
    imgs.append(kitti); masks.append(np.zeros((h, w), np.uint8))
    edge = np.zeros((h, w), np.uint8); edge[:60, :] = 255; edge[:, :50] = 255; edge[-45:, -300:] = 255
    imgs.append(kitti); masks.append(edge)
    n = len(imgs)
    d_i = torch.from_numpy(np.stack(imgs)).cuda(); d_m = torch.from_numpy(np.stack(masks)).cuda()
    det = ORB()
    orc = OracleCvORB()
    for rep in range(2):                       # the second batch reuses the arena: stale planes of the first must not leak
        det.detect_batch_device(d_i.data_ptr(), d_m.data_ptr(), n, w, h)
        culled = 0
        for i in range(n):
            kps, desc = det.batch_fetch(i)
            ko, do = orc.run(imgs[i], masks[i])
            assert len(kps) == len(ko), (i, len(kps), len(ko))
            assert np.array_equal(kps.view(np.uint8), ko.view(np.uint8)), "image %d: keypoints (order included)" % i
            assert np.array_equal(desc, do), "image %d: descriptors" % i
            culled += len(ko) >= 400
        assert culled >= 2                     # some masks are large enough for the quota culls
        d_i = torch.from_numpy(np.stack(imgs[::-1])).cuda(); d_m = torch.from_numpy(np.stack(masks[::-1])).cuda()
        imgs, masks = imgs[::-1], masks[::-1]
    det.close()


def test_randomised_batched_detector_sweep():
    """tools/stress_cvorb_batch.py: random crops / flips of the real frame under random masks, several images per batch, image sizes that
    are no multiple of the kernels' tile / cell sizes - keypoints (order included) and descriptors identical to the CPU restatement"""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_cvorb_batch.py")
    r = subprocess.run([sys.executable, tool, "7", "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical to the CPU restatement" in r.stdout


def test_single_image_call_batched_form_equals_per_level_form(monkeypatch):
    """r06: ps_cvorb_detect_and_compute with a mask is served by the batched device form on a batch of one (PS_CVORB_FAST, default on) and by
    the per-level host-selected form otherwise - also when the batched form's per-level store overflows.  Same keypoints in the same order,
    same descriptors, both ways; the intermediates ps_cvorb_debug_read hands out come from the per-level form in either case."""
    import os
    from PIL import Image
    from pointslot_amd.object_orb import ORB
    seq = sequence.generate(n_frames=1, seed=9)
    img = seq["left"][0]
    small = np.where(seq["seg"][0] != 0, 255, 0).astype(np.uint8)                    # two boxes: served by the batched form
    kitti = np.ascontiguousarray(np.asarray(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_000212_gray.png"))))
    big = np.zeros_like(kitti); big[40:360, 100:1200] = 255                          # most of a textured frame: more FAST keypoints than its stores hold
    for im, mask in ((img, small), (kitti, big)):
        monkeypatch.setenv("PS_CVORB_FAST", "1")
        a = ORB()
        monkeypatch.setenv("PS_CVORB_FAST", "0")
        b = ORB()
        ka, da = a.detectAndCompute(im, mask)
        kb, db = b.detectAndCompute(im, mask)
        assert len(ka) == len(kb) > 30
        assert np.array_equal(ka.view(np.uint8), kb.view(np.uint8)) and np.array_equal(da, db)
        for l in (0, 3, 7):
            assert np.array_equal(a.debug_plane(l, 1), b.debug_plane(l, 1))
            assert np.array_equal(a.debug_fast(l).view(np.uint32), b.debug_fast(l).view(np.uint32))
        # a second call on the same handle (plans and staging buffers are reused)
        k2, d2 = a.detectAndCompute(im, mask)
        assert np.array_equal(k2.view(np.uint8), ka.view(np.uint8)) and np.array_equal(d2, da)
        a.close(); b.close()
