"""The seeded problems behind tests/golden/match_golden.json and opt_golden.json (shared by the generator and the tests)."""
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
