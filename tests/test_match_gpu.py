"""GPU parity of the matcher kernels against the CPU oracle: bit-exact distance matrices and match sets."""
import os
import numpy as np
import pytest

import oracle_lib
from pointslot_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    from pointslot_amd.matcher import ORBmatcher
    m = ORBmatcher(0.9, True)
    yield m
    m.close()


def test_hamming_matrix(matcher):
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (333, 32), dtype=np.uint8); t = rng.integers(0, 256, (1001, 32), dtype=np.uint8)
    t[5] = q[7]; t[6] = ~q[7]
    g = matcher.DescriptorDistanceMatrix(q, t)
    assert np.array_equal(g, oracle_lib.hamming_matrix(q, t))
    assert g[7, 5] == 0 and g[7, 6] == 256


@pytest.mark.parametrize("check_ori", [True, False])
@pytest.mark.parametrize("ratio", [0.9, 0.6])
def test_bruteforce_batch_matches_oracle(check_ori, ratio):
    from pointslot_amd.matcher import ORBmatcher
    m = ORBmatcher(ratio, check_ori)
    shapes = [(300, 320), (1000, 1000), (5, 3), (64, 65), (200, 7), (1, 1), (700, 1500), (33, 4000),
              # the kernels' tier boundaries: register keys up to 256 / 512 train descriptors, train angles in LDS up to 1024
              (40, 256), (40, 257), (70, 512), (40, 513), (20, 1024), (20, 1025)]
    probs = [synth.bruteforce_problem(0x51070010 + k, nq, nt) for k, (nq, nt) in enumerate(shapes)]
    # adversarial: many identical descriptors (exhausts the top-8 lists and forces the rescan path)
    dup = synth.bruteforce_problem(0x51070099, 120, 100)
    dup["t_desc"][:] = dup["t_desc"][0]
    dup["t_desc"][50:, 0] ^= 1
    dup["q_desc"][:] = dup["t_desc"][0]
    dup["q_desc"][::2, 1] ^= 3
    probs.append(dup)
    res = m.SearchByBruceMatching(probs)
    total = 0
    for p, (n, out) in zip(probs, res):
        no, oo = oracle_lib.search_bruteforce(p, ratio, check_ori)
        assert n == no
        assert np.array_equal(out, oo)
        total += n
    assert total > 300
    m.close()


def test_bruteforce_empty_sides(matcher):
    p = synth.bruteforce_problem(5, 10, 10)
    p0 = dict(p); p0["q_valid"] = np.zeros(10, np.uint8)
    (n, out), = matcher.SearchByBruceMatching([p0])
    assert n == 0 and np.all(out == -1)


def _scene_problems(seed, **kw):
    from pointslot_amd.matcher import build_grid
    sc = synth.projection_scene(seed, **kw)
    tr = dict(sc["train"])
    tr["cell_off"], tr["cell_idx"] = build_grid(tr["x"], tr["y"], *tr["grid"])
    base = {"train": tr, "scale_factors": sc["scale_factors"]}
    frame = dict(base, mode="frame", query=sc["frame_query"], tcw=sc["tcw"], tlw=sc["tlw"], K6=sc["K6"], bounds=sc["bounds"], th=sc["th"])
    pts = dict(base, mode="points", query=sc["points_query"], th=1.0)
    pts3 = dict(pts, th=3.0)
    obj = dict(base, mode="points", query=sc["points_query"], th=1.0, object=True)
    return frame, pts, pts3, obj


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_projection_all_variants(check_ori):
    from pointslot_amd.matcher import ORBmatcher
    m = ORBmatcher(0.8, check_ori)
    probs = []
    for seed, kw in ((0x51070020, {}), (0x51070021, {"n": 600, "m": 900, "th": 15.0}), (0x51070022, {"n": 50, "m": 10})):
        probs += list(_scene_problems(seed, **kw))
    backward = dict(probs[0]); backward["tcw"] = np.linalg.inv(probs[0]["tcw"].astype(np.float64)).astype(np.float32)
    mono = dict(probs[0]); mono["mono"] = True
    probs += [backward, mono]
    res = m.SearchByProjection(probs)
    total = 0
    for i, (pr, (n, out)) in enumerate(zip(probs, res)):
        if pr["mode"] == "frame":
            no, oo = oracle_lib.search_projection_frame(pr, check_ori)
        else:
            no, oo = oracle_lib.search_projection_points(pr, 0.8)
        assert n == no, (i, n, no)
        assert np.array_equal(out, oo), (i, int((out != oo).sum()))
        total += n
    assert total > 1500
    m.close()


def _dense_window_problem(seed, n_dense, n_queries=40):
    """a points-mode problem whose queries all project into one spot crowded with n_dense level-0 features: every search window
    holds n_dense candidates (the reference's GetFeaturesInArea has no limit; the first candidate store here holds 256)"""
    from pointslot_amd.matcher import build_grid
    sc = synth.projection_scene(seed, n=n_dense + 300, m=n_queries)
    rng = np.random.default_rng(seed)
    tr = dict(sc["train"])
    for k in ("x", "y", "octave", "u_right", "occupied"):
        tr[k] = np.array(tr[k])
    tr["x"][:n_dense] = (300 + rng.uniform(-3, 3, n_dense)).astype(np.float32)
    tr["y"][:n_dense] = (150 + rng.uniform(-3, 3, n_dense)).astype(np.float32)
    tr["octave"][:n_dense] = 0
    tr["u_right"][:n_dense] = -1
    tr["occupied"][:n_dense] = 0
    tr["cell_off"], tr["cell_idx"] = build_grid(tr["x"], tr["y"], *tr["grid"])
    q = {k: np.array(v) for k, v in sc["points_query"].items()}
    q["proj_x"][:] = 300; q["proj_y"][:] = 150; q["proj_xr"][:] = 250; q["level"][:] = 0; q["view_cos"][:] = 0.9; q["valid"][:] = 1
    # queries look like the dense features: real competition for the same trains, order dependence included
    q["desc"] = np.array(tr["desc"])[rng.integers(0, n_dense, len(q["valid"]))].copy()
    q["desc"][:, 0] ^= rng.integers(0, 4, len(q["valid"])).astype(np.uint8)
    return {"train": tr, "scale_factors": sc["scale_factors"], "mode": "points", "query": q, "th": 1.0}


def test_search_windows_beyond_the_first_candidate_store(matcher):
    """one over-full window used to fail the whole batch.  Now such a problem runs again with the wide store and is
    exact; only a window beyond 1024 candidates fails - that problem alone."""
    frame, pts, _, _ = _scene_problems(0x51070020)
    dense = _dense_window_problem(77, 700)
    res = matcher.SearchByProjection([frame, dense, pts])
    for pr, (n, out) in zip([frame, dense, pts], res):
        no, oo = oracle_lib.search_projection_frame(pr, True) if pr["mode"] == "frame" else oracle_lib.search_projection_points(pr, 0.9)
        assert n == no and np.array_equal(out, oo), (pr["mode"], n, no)
    assert res[1][0] >= 20
    hopeless = _dense_window_problem(78, 1300)
    with pytest.raises(Exception, match="more than 1024 candidates"):
        matcher.SearchByProjection([frame, hopeless])
    res = matcher.SearchByProjection([frame, hopeless, dense], partial_ok=True)
    assert res[1][0] == -1 and np.all(res[1][1] == -1)
    for k in (0, 2):
        pr = [frame, hopeless, dense][k]
        no, oo = oracle_lib.search_projection_frame(pr, True) if pr["mode"] == "frame" else oracle_lib.search_projection_points(pr, 0.9)
        assert res[k][0] == no and np.array_equal(res[k][1], oo)


def test_search_by_projection_with_the_largest_train_set(matcher):
    """32 767 train features in one frame (the most a candidate key's index field holds): pj_resolve stages the train octaves in LDS sized
    by the launch - here the full 32 KB -, the occupancy bitmaps are full length; beside it a small problem in the same call."""
    from pointslot_amd.matcher import build_grid
    sc = synth.projection_scene(0x51070040, n=32767, m=600)
    tr = dict(sc["train"]); tr["cell_off"], tr["cell_idx"] = build_grid(tr["x"], tr["y"], *tr["grid"])
    base = {"train": tr, "scale_factors": sc["scale_factors"]}
    pts = dict(base, mode="points", query=sc["points_query"], th=1.0)
    frame = dict(base, mode="frame", query=sc["frame_query"], tcw=sc["tcw"], tlw=sc["tlw"], K6=sc["K6"], bounds=sc["bounds"], th=sc["th"])
    small = _scene_problems(0x51070022, n=50, m=10)[1]
    res = matcher.SearchByProjection([pts, small, frame])
    for pr, (n, out) in zip([pts, small, frame], res):
        no, oo = oracle_lib.search_projection_frame(pr, True) if pr["mode"] == "frame" else oracle_lib.search_projection_points(pr, 0.9)
        assert n == no and np.array_equal(out, oo), (pr["mode"], n, no)
    assert res[0][0] > 100 and res[2][0] > 100


def test_matchers_with_empty_inputs(matcher):
    from pointslot_amd.matcher import build_grid
    p = synth.bruteforce_problem(3, 0, 12)                      # no queries
    (n, out), = matcher.SearchByBruceMatching([p])
    assert n == 0 and np.all(out == -1) and len(out) == 12
    p = synth.bruteforce_problem(4, 9, 0)                       # no trains
    (n, out), = matcher.SearchByBruceMatching([p])
    assert n == 0 and len(out) == 0
    sc = synth.projection_scene(0x51070030, n=40, m=0)
    tr = dict(sc["train"]); tr["cell_off"], tr["cell_idx"] = build_grid(tr["x"], tr["y"], *tr["grid"])
    pr = {"train": tr, "scale_factors": sc["scale_factors"], "mode": "points", "query": sc["points_query"], "th": 1.0}
    (n, out), = matcher.SearchByProjection([pr])
    assert n == 0 and np.all(out == -1)


def test_distinctive_descriptors(matcher):
    rng = np.random.default_rng(12)
    lists = []
    for n in [1, 2, 3, 7, 50, 64, 65, 128, 0, 31]:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        obs = []
        for _ in range(n):
            bits = np.unpackbits(base)
            flips = rng.integers(0, 256, rng.integers(0, 60))
            bits[flips] ^= 1
            obs.append(np.packbits(bits))
        lists.append(np.array(obs, np.uint8).reshape(-1, 32))
    lists.append(np.tile(rng.integers(0, 256, 32, dtype=np.uint8), (9, 1)))      # all identical: first index wins
    g = matcher.ComputeDistinctiveDescriptors(lists)
    o = oracle_lib.distinctive_descriptors(lists)
    assert np.array_equal(g, o), (g, o)
    assert g[8] == -1 and g[-1] == 0


@pytest.mark.parametrize("seed,kw", [(41, {}), (42, {"th": 5.0}), (43, {"box": (300, 600, 100, 300)}), (44, {"n": 3000, "m": 2500}),
                                     (45, {"m": 1}), (46, {"n": 40, "m": 300})])
def test_fuse_search_bit_exact(matcher, seed, kw):
    """ORBmatcher::Fuse search half (KeyFrame and ObjectKeyFrame variants): best feature and distance per candidate."""
    from pointslot_amd.matcher import build_grid
    pr = synth.fuse_scene(seed, **kw)
    T = pr["train"]
    T["cell_off"], T["cell_idx"] = build_grid(T["x"], T["y"], *T["grid"])
    (gi, gd), = matcher.FuseSearch([pr])
    oi, od = oracle_lib.fuse_search(pr)
    assert np.array_equal(gi, oi), np.nonzero(gi != oi)[0][:10]
    assert np.array_equal(gd, od)


def test_fuse_search_batch(matcher):
    from pointslot_amd.matcher import build_grid
    prs = []
    for s in range(50, 56):
        pr = synth.fuse_scene(s, n=500 + 100 * (s - 50), m=200 + 50 * (s - 50))
        T = pr["train"]
        T["cell_off"], T["cell_idx"] = build_grid(T["x"], T["y"], *T["grid"])
        prs.append(pr)
    res = matcher.FuseSearch(prs)
    for pr, (gi, gd) in zip(prs, res):
        oi, od = oracle_lib.fuse_search(pr)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_randomised_matcher_sweep():
    """many scene sizes / radii / ratios, including small images where thousands of queries fight for a few keypoints (the
    order-dependent resolve then runs its rescan path constantly); every variant must stay bit-exact"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_matchers.py")
    r = subprocess.run([sys.executable, tool, "11", "16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
