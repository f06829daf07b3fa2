"""The CPU checker against the committed matcher / optimiser fixtures (tests/golden/match_golden.json, opt_golden.json, made
by tests/golden/make_match_opt_golden.py): index, mask and count results exactly, FP64 results to 1e-9."""
import hashlib
import json
import os

import numpy as np

import oracle_lib
from golden_cases import matcher_cases, pose_cases, ba_cases, cfse3_cases, fuse_cases, distinctive_case, dynamic_cases, stereo_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_match(g, n, out):
    assert int(n) == g["nmatches"] and int((out >= 0).sum()) == g["assigned"]
    assert sha(out.astype(np.int32)) == g["match_sha256"]
    for j, i in g["first_assignments"]:
        assert out[j] == i


def _check_trace(trace, gt, chi_rtol, strict):
    trace = np.asarray(trace)
    assert len(trace) == len(gt)
    assert np.array_equal(trace[:, 2].astype(int), gt[:, 2].astype(int))          # damping trials per LM iteration
    assert np.allclose(trace[:, 0], gt[:, 0], rtol=chi_rtol, atol=1e-12)           # chi2
    if strict:
        assert np.allclose(trace[:, 1], gt[:, 1], rtol=1e-9, atol=1e-12)           # lambda
    else:
        # lambda follows the gain ratio (chi - chi_new) / scale: once an iteration no longer changes chi2 that ratio is
        # cancellation noise, so a different summation order is only held to lambda where the step was significant
        prev = np.concatenate([[np.inf], gt[:-1, 0]])
        sig = (np.abs(prev - gt[:, 0]) > 1e-4 * gt[:, 0]) & (gt[:, 2] == 1)
        assert not (~np.isclose(trace[:, 1], gt[:, 1], rtol=1e-5) & sig).any()


def check_pose(g, r, tcw, outlier, trace, strict=True):
    """strict: the CPU checker reproduces its own fixture to 1e-9; the GPU path (other summation order) is held to the
    tolerances of tests/test_opt_gpu.py - identical counts, masks and damping trials, chi2 to 1e-9, pose to 1e-6"""
    assert int(r) == g["result"] and int(outlier.sum()) == g["n_outliers"] and sha(outlier.astype(np.uint8)) == g["outlier_sha256"]
    assert np.allclose(np.asarray(tcw, np.float64).reshape(16), g["tcw"], rtol=0, atol=1e-6)
    _check_trace(trace, np.array(g["trace"]), 1e-9, strict)


def check_ba(g, n, poses, pts, erase, trace, strict=True):
    assert int(n) == g["erased"] and sha(np.asarray(erase, np.uint8)) == g["erase_sha256"]
    tol = 1e-9 if strict else 1e-6
    assert np.allclose(poses, g["poses"], rtol=tol, atol=tol)
    assert np.allclose(np.asarray(pts).sum(0), g["points_sum"], rtol=tol, atol=tol * 100) and np.allclose(np.asarray(pts)[:4], g["points_first"], rtol=tol, atol=tol)
    _check_trace(trace, np.array(g["trace"]), 1e-9 if strict else 1e-8, strict)


def test_matcher_fixtures():
    gold = json.load(open(os.path.join(GOLD, "match_golden.json")))
    cases = matcher_cases()
    assert sorted(gold) == sorted(c[0] for c in cases)
    for name, kind, pr, arg in cases:
        if kind == "bruteforce":
            n, out = oracle_lib.search_bruteforce(pr, *arg)
        elif kind == "frame":
            n, out = oracle_lib.search_projection_frame(pr, check_ori=arg)
        else:
            n, out = oracle_lib.search_projection_points(pr, arg)
        check_match(gold[name], n, out)
        assert gold[name]["nmatches"] > 20, name          # the fixtures are not trivially empty


def test_optimiser_fixtures():
    gold = json.load(open(os.path.join(GOLD, "opt_golden.json")))
    for name, p in pose_cases():
        check_pose(gold[name], *oracle_lib.pose_optimize(p, want_trace=True))
    for name, p in ba_cases():
        check_ba(gold[name], *oracle_lib.object_ba(p))


def check_aux(gold, cfse3, fuse, distinctive, dynamic, stereo, strict=True):
    """cfse3: {name: (ok, poses, outliers)}, fuse: {name: (best_idx, best_dist)}, distinctive: indices, dynamic: list of
    (mono_avg, stereo_avg, n_mono, n_stereo), stereo: (kept, u_right, depth)"""
    for name, (ok, poses, outl) in cfse3.items():
        g = gold[name]
        assert int(ok) == g["ok"] and [int(o.sum()) for o in outl] == g["n_outliers"]
        assert [sha(np.asarray(o, np.uint8)) for o in outl] == g["outlier_sha256"]
        assert np.allclose(poses, g["poses"], rtol=0, atol=1e-9 if strict else 1e-6)
    for name, (bi, bd) in fuse.items():
        g = gold[name]
        assert sha(np.asarray(bi, np.int32)) == g["best_idx_sha256"] and sha(np.asarray(bd, np.int32)) == g["best_dist_sha256"]
        assert int((np.asarray(bi) >= 0).sum()) == g["n_found"] > 10
    assert [int(v) for v in distinctive] == gold["distinctive"]["best"]
    for r, g in zip(dynamic, gold["dynamic"]):      # bit-identical on both sides (sorted, sequential FP64 sums)
        assert [float(r[0]).hex(), float(r[1]).hex(), int(r[2]), int(r[3])] == g
    kept, ur, dp = stereo
    g = gold["stereo"]
    assert int(kept) == g["kept"] > 500 and len(ur) == g["n"]
    assert sha(np.asarray(ur, np.float32)) == g["u_right_sha256"] and sha(np.asarray(dp, np.float32)) == g["depth_sha256"]


def test_further_fixtures():
    gold = json.load(open(os.path.join(GOLD, "aux_golden.json")))
    cf = {name: oracle_lib.cfse3_optimize(f["objs"], f["K"]) for name, f in cfse3_cases(oracle_lib.se3_from_mat4f)}
    fu = {name: oracle_lib.fuse_search(pr) for name, pr in fuse_cases()}
    L, R = stereo_case()
    ol, orr = oracle_lib.OracleORB(2000), oracle_lib.OracleORB(2000)
    ol.run(L); orr.run(R)
    st = oracle_lib.stereo_match(ol, orr, np.float32(384.38148 / 721.5377), np.float32(384.38148))
    check_aux(gold, cf, fu, oracle_lib.distinctive_descriptors(distinctive_case()), [oracle_lib.dynamic_discrimination(o) for o in dynamic_cases()], st)
