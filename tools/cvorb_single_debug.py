import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pointslot_amd import sequence
from pointslot_amd.object_orb import ORB
from pointslot_amd.object_tracker import right_mask, object_masks
seq = sequence.generate(n_frames=2, seed=0)
img = seq["left"][1]
mask = sequence.frame_mask(seq, 1)
mr = right_mask(mask)
om_l, om_r = object_masks(mask, mr)
print("mask coverage", (om_l != 0).mean(), (om_r != 0).mean())
det = ORB(1000, 1.2, 8, 19)
for m in (om_l, om_r):
    for rep in range(4):
        t0 = time.perf_counter()
        k, d = det.detectAndCompute(img, m)
        print("n=%d  %.3f ms" % (len(k), (time.perf_counter() - t0) * 1e3))
