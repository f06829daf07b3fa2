"""Quick timing of the pose-only LM kernels (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pointslot_amd import synth
from pointslot_amd.optimizer import Optimizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = [synth.pose_problem(0x51070003 + k) for k in range(n)]
opt = Optimizer()
opt.PoseOptimization(frames[:1])
ms = []
for _ in range(5):
    res = opt.PoseOptimization(frames)
    ms.append(opt.last_kernel_ms())
print("pose_lm %d frames: kernel ms %s  inliers0 %d  checksum %.9f" % (n, [round(m, 3) for m in ms], res[0][0], float(sum(r[1].sum() for r in res))))
