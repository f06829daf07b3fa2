"""Stage timing of the ORB pipeline on a batch made of the committed real KITTI frame (developer tool)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from pointslot_amd.extractor import ORBextractor
img = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "kitti_000212_gray.png")))
n = 128
h, w = img.shape
batch = np.stack([np.roll(img, 3 * k, axis=1) for k in range(n)])          # shifted copies: distinct images, same statistics
d = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
ex = ORBextractor(2000, 1.2, 8, 20, 5, max_batch=n)
for _ in range(3):
    ex.extract_batch_device(d.data_ptr(), n, w, h, w, w * h)
ex.sync()
ex.enable_stage_timing(True)
ex.extract_batch_device(d.data_ptr(), n, w, h, w, w * h)
ex.sync()
st = ex.stage_times()
k, _ = ex.fetch(0)
print("real frame %dx%d x %d: keypoints %d, stage ms:" % (w, h, n, len(k)), {a: round(b, 4) for a, b in st.items()}, "total", round(sum(st.values()), 4))
