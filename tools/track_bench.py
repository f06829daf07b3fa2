"""Developer bench of the device-resident lockstep tracker: S sequences (n_distinct generated ones, repeated), images in HBM.
  python tools/track_bench.py [S] [frames] [groups]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd import sequence  # noqa: E402
from pointslot_amd.tracker_device import LockstepTracker  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nd = 4
seqs = [sequence.generate(n_frames=n, seed=40 + k, step=0.05 + 0.01 * k) for k in range(nd)]
h, w = seqs[0]["left"][0].shape
base = np.stack([np.stack([q["left"], q["right"]], 1) for q in seqs], 1)   # [n, nd, 2, h, w]
Sg = S // G
d = torch.from_numpy(base).cuda()
idx = torch.arange(Sg, device="cuda") % nd
imgs = d[:, idx].contiguous()                                              # [n, Sg, 2, h, w]
trks = [LockstepTracker(Sg, seqs[0]["K"], seqs[0]["bf"], w, h, max_steps=n) for _ in range(G)]
for rep in range(2):
    for t in trks:
        t.reset()
        t.enable_stage_timing(rep == 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        for t in trks:
            t.step_device(imgs[i].data_ptr())
        if i == 1:
            for t in trks:
                t.sync()
            t1 = time.perf_counter()
    for t in trks:
        t.sync()
    dt = time.perf_counter() - t1
print("S=%d groups=%d: %.3f ms per step of %d frames -> %.0f tracked frames/s (steps 2..%d)" % (S, G, dt / (n - 2) * 1e3, S, S * (n - 2) / dt, n - 1))
tcw, st = trks[0].fetch()
print("tracked %d of %d; mm_matches median %d, lm_inliers median %d, retried %d" % (st["tracked"].sum(), st.size, np.median(st["mm_matches"][1:]),
                                                                                    np.median(st["lm_inliers"][1:]), st["retried"].sum()))
err = 0.0
for k in range(min(Sg, nd)):
    twc = np.array([-(tcw[i, k, :3, :3].T @ tcw[i, k, :3, 3]) for i in range(n)])
    err = max(err, float(np.abs(twc - seqs[k]["twc"][:, :, 3]).max()))
print("max position error %.4f m" % err)
for k, v in trks[0].stage_times().items():
    print("  %-28s %.4f ms" % (k, v))
