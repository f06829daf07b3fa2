"""Generates tests/golden/chain_ba_fixture.npz ON THE GPU BOX (the chain runs over the HIP library):
    gpurun -- 'python tests/golden/make_chain_ba_fixture.py gpurun_out/chain_ba_fixture.npz'
then copy the file to tests/golden/.  A 40-frame generated drive goes through the per-frame chain (camera + object half, the per-call
C-ABI); the object with the most object keyframes (one every third frame) gives the ObjectLocalBundleAdjustment graph, its frames the
DynamicStaticDiscrimination problems.  Inputs only: the expected outputs are the CPU checker's, computed by the tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pointslot_amd import sequence  # noqa: E402
from pointslot_amd.tracker import HipBackend  # noqa: E402
import chain_harvest  # noqa: E402

n = 40
seq = sequence.generate_drive(n_frames=n, seed=4, texture=sequence.kitti_texture(), speed=0.6, yaw_rate_deg=0.4)
be = HipBackend()
vo, per_obj, dyn = chain_harvest.run_and_harvest(be, seq, n)
tid, g = chain_harvest.best_graph(per_obj, seq["K"], seq["bf"])
dyn = [d for d in dyn if d["track_id"] == tid][:12]
out = {"track_id": tid, "n_frames": n}
for k, v in g.items():
    out["ba_" + k] = np.asarray(v)
out["n_dyn"] = len(dyn)
for i, d in enumerate(dyn):
    for k, v in d.items():
        out["dyn%d_%s" % (i, k)] = np.asarray(v)
np.savez_compressed(sys.argv[1], **out)
print("object %d: %d keyframes (frames %s), %d points, %d edges; %d discrimination problems" % (
    tid, len(g["poses"]), g["frames"].tolist(), len(g["points"]), len(g["e_pose"]), len(dyn)))
