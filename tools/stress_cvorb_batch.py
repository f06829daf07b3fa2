"""Randomised parity sweep of the batched, device-resident object detector (ps_cvorb_detect_batch_device) against the CPU restatement
of cv::ORB (developer tool, run on the GPU box): the real KITTI frame and generated frames, flipped / cropped to sizes that are no
multiple of the kernels' tile and cell sizes, under random box / ellipse masks of all sizes (empty and near-full ones included), several
images per batch so that the worklists interleave.   python tools/stress_cvorb_batch.py [seed] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from PIL import Image
from oracle_lib import OracleCvORB
from pointslot_amd import sequence
from pointslot_amd.object_orb import ORB

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kitti = np.ascontiguousarray(np.asarray(Image.open(os.path.join(root, "tests", "golden", "kitti_000212_gray.png"))))


def blob_mask(h, w, nblobs, big):
    m = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(nblobs):
        bw, bh = (int(rng.integers(150, 500)), int(rng.integers(80, 220))) if big else (int(rng.integers(12, 160)), int(rng.integers(10, 90)))
        x0, y0 = int(rng.integers(-bw // 2, w - bw // 2)), int(rng.integers(-bh // 2, h - bh // 2))
        if rng.random() < 0.5:
            m[max(y0, 0):y0 + bh, max(x0, 0):x0 + bw] = 255
        else:
            m[((xx - x0 - bw / 2) / (bw / 2)) ** 2 + ((yy - y0 - bh / 2) / (bh / 2)) ** 2 <= 1] = 255
    return m


total = culled = 0
t0 = time.time()
for b in range(nbatch):
    # one image size per batch (the batched call takes images of one size)
    h = int(rng.integers(200, kitti.shape[0] + 1)); w = int(rng.integers(400, kitti.shape[1] + 1))
    y0 = int(rng.integers(0, kitti.shape[0] - h + 1)); x0 = int(rng.integers(0, kitti.shape[1] - w + 1))
    n = int(rng.integers(2, 7))
    imgs, masks = [], []
    for i in range(n):
        im = kitti[y0:y0 + h, x0:x0 + w]
        if rng.random() < 0.5: im = im[:, ::-1]
        if rng.random() < 0.3: im = im[::-1]
        imgs.append(np.ascontiguousarray(im))
        kind = rng.random()
        masks.append(np.zeros((h, w), np.uint8) if kind < 0.08 else blob_mask(h, w, int(rng.integers(1, 5)), big=kind > 0.6))
    d_i = torch.from_numpy(np.stack(imgs)).cuda(); d_m = torch.from_numpy(np.stack(masks)).cuda()
    det = ORB(); orc = OracleCvORB()
    det.detect_batch_device(d_i.data_ptr(), d_m.data_ptr(), n, w, h)
    for i in range(n):
        try:
            kps, desc = det.batch_fetch(i)
        except Exception as e:
            if "more than" in str(e):     # a per-level candidate list beyond the detector's capacity: reported, not wrong
                print("batch %d image %d: %s" % (b, i, e)); continue
            raise
        ko, do = orc.run(imgs[i], masks[i])
        assert len(kps) == len(ko), (b, i, w, h, len(kps), len(ko))
        assert np.array_equal(kps.view(np.uint8), ko.view(np.uint8)), "batch %d image %d (%d x %d): keypoints (order included)" % (b, i, w, h)
        assert np.array_equal(desc, do), "batch %d image %d (%d x %d): descriptors" % (b, i, w, h)
        total += 1; culled += len(ko) >= 400
    det.close()
print("stress_cvorb_batch seed %d: %d images in %d batches identical to the CPU restatement (%d with quota culls), %.0f s" % (seed, total, nbatch, culled, time.time() - t0))
