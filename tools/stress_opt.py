"""Randomised parity sweep of the optimiser kernels against the CPU checker (developer tool, run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pointslot_amd._lib import poison_lds
import oracle_lib
from pointslot_amd import synth
from pointslot_amd.optimizer import Optimizer

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
opt = Optimizer()
poison_lds(0xFFFFFFFF)      # uninitialised-LDS reads become deterministic failures
bad = 0
frames = []
for it in range(ncase):
    frames.append(synth.pose_problem(int(rng.integers(1, 1 << 30)), n=int(rng.choice([20, 100, 700, 2000, 3000])), outlier_frac=float(rng.choice([0.0, 0.1, 0.3, 0.5])),
                                     noise=float(rng.choice([0.5, 1.0, 2.0])), mono_frac=float(rng.choice([0.0, 0.3, 1.0])), valid_frac=float(rng.choice([1.0, 0.6]))))
res = opt.PoseOptimization(frames)
worst = 0.0
for it, (f, (n, tcw, out)) in enumerate(zip(frames, res)):
    no, to, oo, _ = oracle_lib.pose_optimize(f)
    d = float(np.abs(tcw.astype(np.float64) - to).max())
    worst = max(worst, d)
    if n != no or not np.array_equal(out, oo) or d > 1e-5:
        bad += 1
        print("MISMATCH pose case %d: inliers %d vs %d, outlier flags differ %d, max |dT| %.2e" % (it, n, no, int((out != oo).sum()), d))
print("pose sweep: %d cases, worst |dT| %.2e" % (ncase, worst))
nba = max(2, ncase // 10)
graphs = [synth.object_ba_problem(int(rng.integers(1, 1 << 30)), n_kf=int(rng.choice([5, 20, 50])), n_pts=int(rng.choice([40, 300])), p_vis=float(rng.choice([0.6, 1.0])),
                                  perturb=(0.05, 1.0, 0.02), perturb_axis="z") for _ in range(nba)]
r = opt.ObjectLocalBundleAdjustment(graphs)
for it, (g, x) in enumerate(zip(graphs, r)):
    o = oracle_lib.object_ba(g)
    er_o = o[3] if isinstance(o, tuple) else None
    same_erase = np.array_equal(np.asarray(x["erase"]), np.asarray(o[3])) if er_o is not None else True
    dp = float(np.abs(np.asarray(x["poses"]) - np.asarray(o[1])).max())
    if not same_erase or dp > 1e-5:
        bad += 1
        print("MISMATCH BA case %d: erase equal %s, max pose diff %.2e" % (it, same_erase, dp))
print("opt stress: %d mismatches" % bad)
opt.close()
sys.exit(1 if bad else 0)
