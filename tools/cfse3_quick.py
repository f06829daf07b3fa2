"""Developer tool: CFSE3ObjStateOptimization on 64 frames x k objects x 150 points (the a15 bench leg's problems); prints kernel ms.
With a -DPS_PO_PROFILE build (PS_LIB_PATH) the kernel prints the phase ticks of problem 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pointslot_amd import synth
from pointslot_amd.optimizer import Optimizer, se3_from_mat4f
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
opt = Optimizer()


def cf_frame(seed, k):
    rng = np.random.default_rng(seed)
    objs = []
    for j in range(k):
        pp = synth.pose_problem(seed * 100 + j, n=150, outlier_frac=0.15, mono_frac=0.2, valid_frac=0.8)
        Tp = pp["tcw_true"].copy(); Tp[:3, 3] += rng.uniform(-0.2, 0.2, 3)
        objs.append({"xo": pp["xw"], "obs": pp["obs"], "inv_sigma2": pp["inv_sigma2"], "valid": pp["valid"], "pose7": se3_from_mat4f(Tp.astype(np.float32))})
    return {"objs": objs, "K": pp["K"]}


frames = [cf_frame(50 + i, k) for i in range(64)]
opt.CFSE3ObjStateOptimization(frames[:2])
for _ in range(3):
    opt.CFSE3ObjStateOptimization(frames)
    print("cfse3 64 frames x %d objects: kernel %.3f ms" % (k, opt.last_kernel_ms()))
