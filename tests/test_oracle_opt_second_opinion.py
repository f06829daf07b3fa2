"""Second opinions for the pose-only optimisers: tests/opt_second_opinion.py - numpy, own residuals / Jacobians / SE3 /
6 x 6 solves, no code shared with oracle/ - against liboracle.so: the whole 4 x 10 schedule with re-classification on the committed golden
cases and on all 64 frames of BASELINE config 3, CFSE3 with its numeric prior Jacobian on the golden graphs.  CPU only."""
import numpy as np
import pytest

import opt_second_opinion as so
import oracle_lib
from golden_cases import cfse3_cases, pose_cases
from pointslot_amd import synth


def _pose_err(Ta, Tb):
    """|log(Ta^-1 Tb)| (rotation angle + translation distance) of two float 4 x 4 poses"""
    D = np.linalg.inv(Ta.astype(np.float64)) @ Tb.astype(np.float64)
    ang = np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1))
    return float(ang + np.linalg.norm(D[:3, 3]))


def _compare_pose(p, name):
    r, T, out, tr = so.pose_optimization(p)
    ro, To, oo, tro = oracle_lib.pose_optimize(p, want_trace=True)
    assert r == ro, (name, r, ro)
    assert np.array_equal(out, oo), (name, int((out != oo).sum()))
    assert _pose_err(T, To) <= 1e-6, (name, _pose_err(T, To))
    assert len(tr) == len(tro), (name, len(tr), len(tro))
    for k, ((c, lam, q), (co, lo, qo)) in enumerate(zip(tr, tro)):
        assert abs(c - co) <= 1e-9 * max(1.0, abs(co)), (name, k, c, co)
        # The gain ratio is (chi2 before - chi2 after) / scale: once an iteration moves chi2 by a fraction d of itself, the difference
        # carries the sums' rounding (~1e-13 chi2 over 2000 edges, in a different order here) amplified by 1 / d - and lambda is a function
        # of it.  lambda is therefore compared to 1e-6, widened by that amplification; an iteration that no longer moves chi2 at all
        # (d < 1e-9) has a gain ratio made of rounding noise: its trial count may differ too.  Everywhere else both are the same.
        d = abs(co - tro[k - 1][0]) / max(1.0, abs(co)) if k > 0 else 1.0
        if d > 1e-9:
            assert q == int(qo) and abs(lam - lo) <= max(1e-6, 1e-11 / d) * abs(lo), (name, k, q, qo, lam, lo, d)
    return r, tr


@pytest.mark.parametrize("name,p", pose_cases())
def test_pose_optimization_schedule_equals_the_restatement_on_the_golden_cases(name, p):
    r, tr = _compare_pose(p, name)
    assert r > 0.7 * int(np.asarray(p["valid"]).sum()) and len(tr) >= 8


def test_pose_optimization_all_64_frames_of_config_3():
    """BASELINE configs[2]: 64 frames x 2000 stereo edges, 10 % outliers, the exact 4 x 10 schedule: return values and outlier masks
    identical, chi2 traces to 1e-9, poses to 1e-6"""
    tot = 0
    for k in range(64):
        r, tr = _compare_pose(synth.pose_problem(0x51070003 + k), "frame %d" % k)
        tot += len(tr)
    assert tot > 64 * 10


def test_pose_optimization_edge_cases():
    # fewer than 15 correspondences: 0, nothing touched
    p = synth.pose_problem(7, n=14)
    r, T, out, tr = so.pose_optimization(p)
    ro, To, oo, _ = oracle_lib.pose_optimize(p)
    assert r == ro == 0 and np.array_equal(T, To) and np.array_equal(out, oo)
    # monocular and stereo mixed, invalid slots, heavy outliers
    for seed, kw in ((21, dict(n=300, mono_frac=1.0)), (22, dict(n=600, outlier_frac=0.45, mono_frac=0.3, valid_frac=0.5)), (23, dict(n=40, noise=3.0))):
        _compare_pose(synth.pose_problem(seed, **kw), "seed %d" % seed)


def _compare_cfse3(frame, name):
    ok, poses, outs = so.cfse3_optimization(frame["objs"], frame["K"])
    oko, poseso, outso = oracle_lib.cfse3_optimize(frame["objs"], frame["K"])
    assert ok == oko, name
    # (The prior edge's Jacobian is a central difference with delta = 1e-9 through exp / quaternion product / normalisation - a recipe for
    # 1e-7 of rounding noise per entry.  Measured: the two implementations end within 1e-13 of each other on these graphs; the translation
    # rows of that Jacobian are dominated by exact terms.)
    assert np.abs(poses - poseso).max() <= 1e-9, (name, np.abs(poses - poseso).max())
    for o, (a, b) in enumerate(zip(outs, outso)):
        assert np.array_equal(a, b), (name, o, int((a != b).sum()))
    return poses


def test_cfse3_equals_the_restatement_on_the_golden_graphs():
    for name, frame in cfse3_cases(lambda T: np.concatenate([so.se3_from_mat4f(T)[1], so.se3_from_mat4f(T)[0]])):
        poses = _compare_cfse3(frame, name)
        # ... and the estimates moved towards the generating poses, away from the perturbed start
        for o, obj in enumerate(frame["objs"]):
            assert np.linalg.norm(poses[o][:3] - obj["pose7"][:3]) > 1e-3


def test_cfse3_small_and_empty_graphs():
    name, frame = cfse3_cases(lambda T: np.concatenate([so.se3_from_mat4f(T)[1], so.se3_from_mat4f(T)[0]]))[0]
    small = {"K": frame["K"], "objs": [dict(frame["objs"][0], valid=np.concatenate([np.ones(10, np.uint8), np.zeros(len(frame["objs"][0]["valid"]) - 10, np.uint8)]))]}
    ok, poses, outs = so.cfse3_optimization(small["objs"], small["K"])
    oko, poseso, outso = oracle_lib.cfse3_optimize(small["objs"], small["K"])
    assert ok == oko == 0 and np.array_equal(poses, poseso)          # 10 + 1 edges < 15: false, nothing moved
    assert so.cfse3_optimization([], frame["K"])[0] == 0


def test_se3_conversions_agree_with_the_restatement():
    rng = np.random.default_rng(3)
    for _ in range(20):
        u = rng.normal(size=6) * np.array([0.3, 0.3, 0.3, 2, 2, 2])
        q, t = so.se3_exp(u)
        p7 = oracle_lib.se3_exp(u)
        mine = np.concatenate([t, so.normalize_rotation(q)])
        assert np.abs(mine - p7).max() < 1e-12
        T = so.se3_to_mat4f((so.normalize_rotation(q), t))
        assert np.array_equal(T, oracle_lib.se3_to_mat4f(p7))
        back = so.se3_from_mat4f(T)
        assert np.abs(np.concatenate([back[1], so.normalize_rotation(back[0])]) - oracle_lib.se3_from_mat4f(T)).max() < 1e-12


# ---- ObjectLocalBundleAdjustment (a16 / a17 / a19): tests/ba_second_opinion.py, dense solves of the full normal equations -------------------
def _compare_ba(p, name):
    import ba_second_opinion as bso
    n, poses, pts, erase, tr = bso.object_local_bundle_adjustment(p)
    no, poseso, ptso, eraseo, tro = oracle_lib.object_ba(p)
    assert len(tr) == len(tro), (name, len(tr), len(tro))
    for k, ((c, lam, q), (co, lo, qo)) in enumerate(zip(tr, tro)):
        assert abs(c - co) <= 1e-8 * max(1.0, abs(co)), (name, k, c, co)
        d = abs(co - tro[k - 1][0]) / max(1.0, abs(co)) if k > 0 else 1.0
        if d > 1e-9:      # (see _compare_pose: lambda and the trial count of an iteration that still moves chi2)
            assert q == int(qo) and abs(lam - lo) <= max(1e-6, 1e-10 / d) * abs(lo), (name, k, q, qo, lam, lo, d)
    assert n == no and np.array_equal(erase, eraseo), (name, n, no, int((erase != eraseo).sum()))
    # poses as (t, q): q and -q are the same rotation, both sides keep w >= 0
    assert np.abs(poses - poseso).max() <= 1e-6, (name, np.abs(poses - poseso).max())
    assert np.abs(pts - ptso).max() <= 1e-6 * max(1.0, np.abs(ptso).max()), name
    return n, tr


def test_object_ba_schedule_equals_the_restatement_on_the_golden_graphs():
    """Schur LM of the restatement against a dense LM on the full normal equations with its own edges: 5 robust iterations, the chi2 / depth pass,
    10 plain iterations, erase list - every iteration's chi2 to 1e-8, lambda, trial counts, erase lists identical, poses and points to 1e-6"""
    from golden_cases import ba_cases
    for name, p in ba_cases():
        n, tr = _compare_ba(p, name)
        assert len(tr) >= 6 and n > 0


def test_object_ba_with_monocular_edges_fixed_extra_cameras_and_full_se3_poses():
    p = synth.object_ba_problem(0x51070044, n_kf=5, n_pts=14, p_vis=0.8, outlier_frac=0.1, mono_frac=0.3, perturb=(0.1, 2.0, 0.05), perturb_axis="z", n_fixed_extra=1)
    _compare_ba(p, "mono + fixed extra")
    # LocalBundleAdjustment's vertices: plain VertexSE3Expmap poses - the roll / pitch lock off
    q = synth.object_ba_problem(0x51070045, n_kf=6, n_pts=30, p_vis=0.7, perturb=(0.05, 1.0, 0.02), perturb_axis="z")
    q["pose_flags"] = (np.asarray(q["pose_flags"]) & 1).astype(np.uint8)
    _compare_ba(q, "full SE3")


def test_norollpitch_update_against_the_restatement():
    import ba_second_opinion as bso
    rng = np.random.default_rng(11)
    for _ in range(20):
        u = rng.normal(size=6) * np.array([0.0, 0.0, 0.4, 1, 1, 1])
        q, t = bso.exptwist_norollpitch(u)
        p7 = oracle_lib.se3_exp(u, True)
        assert np.abs(np.concatenate([t, so.normalize_rotation(q)]) - p7).max() < 1e-12
