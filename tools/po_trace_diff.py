"""Developer tool: per-iteration traces (chi2, lambda, trials) of PoseOptimization on config 3 + the test's edge cases, GPU vs the CPU checker."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from pointslot_amd import synth
from pointslot_amd.optimizer import Optimizer
frames = [synth.pose_problem(0x51070003 + k) for k in range(64)]
frames.append(synth.pose_problem(77, n=400, mono_frac=0.5, valid_frac=0.7))
frames.append(synth.pose_problem(78, n=14))
frames.append(synth.pose_problem(79, n=200, outlier_frac=0.6))
frames.append(synth.pose_problem(80, n=3000, noise=3.0))
frames[-1]["outlier0"] = (np.arange(3000) % 7 == 0).astype(np.uint8)
frames.append(synth.pose_problem(81, n=120))
frames[-1]["inv_sigma2"] = np.zeros_like(frames[-1]["inv_sigma2"])
opt = Optimizer()
opt.enable_trace(True)
res = opt.PoseOptimization(frames)
nd = 0
for i, f in enumerate(frames):
    ro, to, oo, tro = oracle_lib.pose_optimize(f, True)
    trg = opt.get_trace(i)
    same = len(trg) == len(tro) and np.array_equal(trg[:, 2], tro[:, 2])
    if not same or res[i][0] != ro or not np.array_equal(res[i][2], oo):
        nd += 1
        print("frame %d: r %d/%d mask diff %d" % (i, res[i][0], ro, int((res[i][2] != oo).sum())))
        print("  gpu:", " | ".join("%.9g %.3g %d" % tuple(r) for r in trg))
        print("  cpu:", " | ".join("%.9g %.3g %d" % tuple(r) for r in tro))
print("%d of %d frames differ in trial counts / result" % (nd, len(frames)))
