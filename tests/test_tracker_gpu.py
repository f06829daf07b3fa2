"""End-to-end parity of the chained hot path: the GPU backend and the CPU checker drive the same
tracking loop over the same generated sequence; extraction, stereo and matching are bit-exact, so the two runs see the same
matches frame after frame and the trajectories agree to the optimiser's tolerance."""
import numpy as np
import pytest

from pointslot_amd import sequence
from pointslot_amd.tracker import HipBackend, StereoOdometry
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


def run(seq, backend, **kw):
    h, w = seq["left"][0].shape
    vo = StereoOdometry(backend, seq["K"], seq["bf"], w, h, **kw)
    for l, r in zip(seq["left"], seq["right"]):
        vo.track(l, r)
    return vo


@pytest.mark.parametrize("shape,frames,local_map", [((800, 300), 8, True), ((1242, 375), 5, True), ((1242, 375), 4, False)])
def test_trajectory_parity(shape, frames, local_map):
    seq = sequence.generate(n_frames=frames, seed=4, w=shape[0], h=shape[1])
    be = HipBackend()
    g = run(seq, be, track_local_map=local_map)
    o = run(seq, OracleBackend(), track_local_map=local_map)
    be.close()
    assert g.state == o.state == "OK"
    assert g.stats == o.stats, (g.stats, o.stats)            # keypoint / stereo / match / inlier counts of every frame
    for k, (a, b) in enumerate(zip(g.trajectory, o.trajectory)):
        assert np.abs(a - b).max() < 2e-5, (k, np.abs(a - b).max())           # float32 4x4 poses, FP64 LM to 1e-6
    twc = np.array([-(t[:3, :3].T @ t[:3, 3]) for t in g.trajectory])
    assert np.abs(twc - seq["twc"][:, :, 3]).max() < 0.04
