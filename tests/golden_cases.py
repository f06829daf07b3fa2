"""The seeded problems behind tests/golden/match_golden.json, opt_golden.json and aux_golden.json (shared by the generator and
the tests)."""
import numpy as np

from pointslot_amd import synth
from pointslot_amd.matcher import build_grid


def _projection(seed, **kw):
    sc = synth.projection_scene(seed, **kw)
    tr = dict(sc["train"])
    tr["cell_off"], tr["cell_idx"] = build_grid(tr["x"], tr["y"], *tr["grid"])
    base = {"train": tr, "scale_factors": sc["scale_factors"]}
    frame = dict(base, mode="frame", query=sc["frame_query"], tcw=sc["tcw"], tlw=sc["tlw"], K6=sc["K6"], bounds=sc["bounds"], th=sc["th"])
    pts = dict(base, mode="points", query=sc["points_query"], th=1.0)
    return frame, pts


def matcher_cases():
    """(name, kind, problem, argument): kind bruteforce -> (nnratio, check_ori); frame -> check_ori; points -> nnratio"""
    out = []
    for k in range(3):
        out.append(("bruteforce_%d" % k, "bruteforce", synth.bruteforce_problem(0x51070010 + k), (0.9, True)))
    out.append(("bruteforce_no_orientation", "bruteforce", synth.bruteforce_problem(0x51070013, nq=500, nt=480), (0.8, False)))
    for k in range(2):
        frame, pts = _projection(0x51070020 + k)
        out.append(("projection_frame_%d" % k, "frame", frame, True))
        out.append(("projection_points_%d" % k, "points", pts, 0.8))
    frame, pts = _projection(0x51070022, object_mode=True)
    out.append(("projection_object", "points", dict(pts, object=True), 0.9))
    frame, _ = _projection(0x51070023, th=14.0)
    out.append(("projection_frame_wide_no_orientation", "frame", frame, False))
    return out


def pose_cases():
    return [("pose_%d" % k, synth.pose_problem(0x51070003 + k)) for k in range(3)] + \
           [("pose_mono_mix", synth.pose_problem(0x51070040, n=800, mono_frac=0.4, valid_frac=0.8))]


def ba_cases():
    return [("object_ba_small", synth.object_ba_problem(0x51070050, n_kf=8, n_pts=40, perturb=(0.05, 1.0, 0.02), perturb_axis="z")),
            ("object_ba_sparse", synth.object_ba_problem(0x51070051, n_kf=12, n_pts=60, p_vis=0.6, perturb=(0.05, 1.0, 0.02), perturb_axis="z"))]


def cfse3_cases(se3_from_mat4f):
    """frames of CFSE3ObjStateOptimization: object-frame points seen through a perturbed object pose (se3_from_mat4f: the
    float matrix -> 7-double pose conversion of whichever side builds the case; both sides' are bit-identical)"""
    out = []
    for seed, k in ((11, 1), (12, 3)):
        rng = np.random.default_rng(seed)
        objs = []
        for o in range(k):
            p = synth.pose_problem(seed * 100 + o, n=120 + 30 * o, outlier_frac=0.15, mono_frac=0.2, valid_frac=0.8)
            Tp = p["tcw_true"].copy(); Tp[:3, 3] += rng.uniform(-0.2, 0.2, 3)
            objs.append({"xo": p["xw"], "obs": p["obs"], "inv_sigma2": p["inv_sigma2"], "valid": p["valid"], "pose7": se3_from_mat4f(Tp.astype(np.float32))})
        out.append(("cfse3_%d_objects" % k, {"objs": objs, "K": p["K"]}))
    return out


def fuse_cases():
    out = []
    for seed, kw in ((71, {}), (72, {"box": (300, 600, 100, 300)})):
        pr = synth.fuse_scene(seed, **kw)
        T = pr["train"]
        T["cell_off"], T["cell_idx"] = build_grid(T["x"], T["y"], *T["grid"])
        out.append(("fuse_%d" % seed, pr))
    return out


def distinctive_case():
    rng = np.random.default_rng(12)
    lists = []
    for n in [1, 2, 3, 7, 50, 64, 65, 128, 31]:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        obs = []
        for _ in range(n):
            d = base.copy()
            for b in rng.integers(0, 256, rng.integers(0, 40)):
                d[b >> 3] ^= np.uint8(1 << (b & 7))
            obs.append(d)
        lists.append(np.array(obs, np.uint8).reshape(n, 32))
    return lists


def dynamic_cases():
    return [synth.dynamic_object(200 + k, n=[300, 40, 1200][k], moving=[0.0, 0.5, 0.2][k], mono_frac=[0.3, 0.0, 0.5][k]) for k in range(3)]


def stereo_case():
    return synth.stereo_pair()
