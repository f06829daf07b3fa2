"""Developer bench of BASELINE configs[4]'s per-GPU share: ONE generated SLOT.MODE-4 drive through the device-resident chain
(ps_tracker_step_slot_device, camera + object chain) with one frame in flight - the latency of the chain, not its throughput.
  python tools/s1_chain_bench.py [frames] [objects 0|1] [sync 0|1]
Under tools/kstat_py.sh the kernel trace of this run gives the per-kernel durations of the S = 1 step; tools/trace_gaps.py
gives the idle time between them."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd import sequence  # noqa: E402
from pointslot_amd.tracker_device import LockstepTracker, pack_detections  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
objects = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
sync_each = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
K = 8
q = sequence.generate_drive(n_frames=n, seed=40, texture=sequence.kitti_texture())
h, w = q["left"][0].shape
imgs = torch.from_numpy(np.stack([q["left"], q["right"]], 1)).cuda()                                   # [n, 2, h, w]
masks = torch.from_numpy(np.stack([sequence.frame_mask(q, i) for i in range(n)])).cuda()
dets = torch.from_numpy(np.stack([pack_detections([sequence.frame_detections(q, i)], K) for i in range(n)]).view(np.uint8)).cuda()
trk = LockstepTracker(1, q["K"], q["bf"], w, h, max_steps=n, max_objects=K if objects else 0)
times = []
for rep in range(2):
    trk.reset()
    torch.cuda.synchronize()
    times = []
    for i in range(n):
        t0 = time.perf_counter()
        if objects:
            trk.step_slot_device(imgs[i].data_ptr(), masks[i].data_ptr(), dets[i].data_ptr())
        else:
            trk.step_device(imgs[i].data_ptr())
        if sync_each:
            trk.sync()
        times.append(time.perf_counter() - t0)
    trk.sync()
t = np.array(times[2:]) * 1e3
tcw, st = trk.fetch()
print("S=1 %s chain, %d frames, %s: median %.3f ms per frame, mean %.3f, min %.3f, max %.3f; tracked %d of %d"
      % ("camera + object" if objects else "camera", n, "one frame in flight" if sync_each else "queued", np.median(t), t.mean(), t.min(), t.max(), int(st["tracked"].sum()), n))
if objects:
    o = trk.fetch_objects()
    print("objects: detections %d, tracked %d, track_ok %d" % (int((o["id"] >= 0).sum()), int((o["tracked"] != 0).sum()), int((o["track_ok"] != 0).sum())))
trk.close()
