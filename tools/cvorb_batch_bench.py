"""Times the batched device-resident object detector (ps_cvorb_detect_batch_device) on the frames the bench tracks: nimg images
(left / right of generated sequences with their object masks) per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pointslot_amd import sequence
from pointslot_amd.object_orb import ORB
from pointslot_amd.object_tracker import right_mask, object_masks

nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seqs = [sequence.generate(n_frames=2, seed=40 + k, texture=sequence.kitti_texture()) for k in range(4)]
imgs, masks = [], []
for i in range(nimg // 2):
    q = seqs[i % 4]
    m = sequence.frame_mask(q, 1)
    ol, orr = object_masks(m, right_mask(m))
    imgs += [q["left"][1], q["right"][1]]; masks += [ol, orr]
h, w = imgs[0].shape
d_i = torch.from_numpy(np.stack(imgs)).cuda(); d_m = torch.from_numpy(np.stack(masks)).cuda()
det = ORB()
for _ in range(3):
    det.detect_batch_device(d_i.data_ptr(), d_m.data_ptr(), nimg, w, h)
torch.cuda.synchronize()
kps, _ = det.batch_fetch(0)
t0 = time.perf_counter()
for _ in range(10):
    det.detect_batch_device(d_i.data_ptr(), d_m.data_ptr(), nimg, w, h)
kps, _ = det.batch_fetch(0)
dt = (time.perf_counter() - t0) / 10
print("cvorb batch: %d images, %.3f ms per call, %d keypoints in image 0, mask coverage %.3f" % (nimg, dt * 1e3, len(kps), float((np.stack(masks[:8]) != 0).mean())))

import ctypes
from pointslot_amd._lib import lib
cnt = (ctypes.c_int32 * 24)(); tiles = (ctypes.c_int32 * 8)()
if lib.psi_cvorb_batch_worklist_counts(det._h, cnt, tiles) == 0:
    for w, name in enumerate(("planes", "FAST", "blur")):
        print("worklist %-6s tiles per image and level: %s  (of %s)" % (name, [round(cnt[w * 8 + l] / nimg, 1) for l in range(8)], list(tiles)))
