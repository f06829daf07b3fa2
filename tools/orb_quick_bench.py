"""Quick stage timing of the ORB pipeline on the GPU (developer tool, not the contract bench)."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd import synth
from pointslot_amd.extractor import ORBextractor

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = synth.stereo_batch(npairs)
n, h, w = batch.shape
d = torch.from_numpy(batch).cuda()
ex = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=n)
for _ in range(3):
    ex.extract_batch_device(d.data_ptr(), n, w, h, w, w * h)
ex.sync()
ex.enable_stage_timing(True)
ex.extract_batch_device(d.data_ptr(), n, w, h, w, w * h)
ex.sync()
st = ex.stage_times()
print("batch of %d images: stage ms:" % n, {k: round(v, 4) for k, v in st.items()}, "total", round(sum(st.values()), 4))
ex.enable_stage_timing(False)
t = time.time()
K = 20
for _ in range(K):
    ex.extract_batch_device(d.data_ptr(), n, w, h, w, w * h)
ex.sync()
dt = (time.time() - t) / K
print("wall per batch %.3f ms -> %.1f stereo frames/s" % (dt * 1e3, npairs / dt))
