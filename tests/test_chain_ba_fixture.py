"""The back end takes what the chain produces: the ObjectLocalBundleAdjustment graph and the DynamicStaticDiscrimination
problems harvested from a generated 40-frame drive (tests/golden/chain_ba_fixture.npz, made on the GPU box by
tests/golden/make_chain_ba_fixture.py: object keyframes every third frame, the reference's hand-off Tracking.cc:1475-1477 ->
Optimizer.cc:755-818) through ps_object_ba_batch / ps_dynamic_discrimination_batch against the CPU checker - and a shorter drive
harvested live on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle_lib  # noqa: E402


def _fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "chain_ba_fixture.npz"))
    g = {k[3:]: z[k] for k in z.files if k.startswith("ba_")}
    g["K"] = tuple(np.float32(v) for v in g["K"])
    dyn = []
    for i in range(int(z["n_dyn"])):
        d = {k[len("dyn%d_" % i):]: z[k] for k in z.files if k.startswith("dyn%d_" % i)}
        d["K"] = tuple(np.float32(v) for v in d["K"]); d["mbf"] = np.float32(d["mbf"])
        dyn.append(d)
    return g, dyn


def test_the_harvested_graph_has_the_reference_shape_and_the_checker_optimises_it():
    g, dyn = _fixture()
    P, L, E = len(g["poses"]), len(g["points"]), len(g["e_pose"])
    assert P == 13 and 80 <= L <= 400 and E >= 5 * P                     # a dozen keyframes, ~100+ points
    assert g["pose_flags"][0] == 3 and (g["pose_flags"][1:] == 2).all()  # the first keyframe fixed, the others VertexSE3Fix
    assert np.bincount(g["e_point"], minlength=L).min() >= 2             # every point seen by two keyframes
    n, po, pt, er, tr = oracle_lib.object_ba(g)
    assert len(tr) >= 2 and tr[-1, 0] <= tr[0, 0]                        # chi2 does not grow over the schedule
    assert n < 0.5 * E                                                   # the chain's inlier observations mostly survive the chi2 pass
    assert np.abs(po - g["poses"]).max() < 0.5                           # ... and the tracked poses were already near the optimum
    assert len(dyn) == 12
    for d in dyn:
        e = oracle_lib.dynamic_discrimination(d)
        assert np.isfinite(e[0]) and np.isfinite(e[1])


@pytest.mark.gpu
def test_object_ba_and_discrimination_on_the_harvested_inputs():
    from pointslot_amd.optimizer import Optimizer
    from test_opt_gpu import _ba_check
    g, dyn = _fixture()
    opt = Optimizer()
    r = opt.ObjectLocalBundleAdjustment([g, g])          # (twice in one batch: the problems are independent)
    _ba_check(g, r[0], "harvested graph")
    _ba_check(g, r[1], "harvested graph, second copy")
    res = opt.DynamicStaticDiscrimination(dyn)
    for d, x in zip(dyn, res):
        e = oracle_lib.dynamic_discrimination(d)
        assert x[2] == e[2] and x[3] == e[3], (x, e)
        assert np.float64(x[0]).tobytes() == np.float64(e[0]).tobytes() and np.float64(x[1]).tobytes() == np.float64(e[1]).tobytes(), (x, e)
    opt.close()


@pytest.mark.gpu
def test_a_live_harvest_goes_through_the_back_end():
    """16 frames through the chain on the GPU right now, harvested, optimised on the GPU and by the checker"""
    from pointslot_amd import sequence
    from pointslot_amd.optimizer import Optimizer
    from pointslot_amd.tracker import HipBackend
    from test_opt_gpu import _ba_check
    import chain_harvest
    n = 16
    seq = sequence.generate_drive(n_frames=n, seed=11, texture=sequence.kitti_texture(), speed=0.6, yaw_rate_deg=0.4)
    be = HipBackend()
    vo, per_obj, dyn = chain_harvest.run_and_harvest(be, seq, n, kf_every=2)
    be.close()
    assert per_obj and dyn
    tid, g = chain_harvest.best_graph(per_obj, seq["K"], seq["bf"])
    assert len(g["poses"]) >= 5 and len(g["points"]) >= 40
    opt = Optimizer()
    r = opt.ObjectLocalBundleAdjustment([g])
    _ba_check(g, r[0], "live graph")
    res = opt.DynamicStaticDiscrimination(dyn[:8])
    for d, x in zip(dyn[:8], res):
        e = oracle_lib.dynamic_discrimination(d)
        assert x[2] == e[2] and x[3] == e[3]
        assert np.float64(x[0]).tobytes() == np.float64(e[0]).tobytes() and np.float64(x[1]).tobytes() == np.float64(e[1]).tobytes()
    opt.close()
