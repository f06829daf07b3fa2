"""GPU parity of the optimiser kernels against the CPU oracle.  FP64 sums are reduced in a different order on the
GPU, so poses are a TOLERANCE target: || log(T_gpu^-1 T_cpu) || <= 1e-6 (BASELINE.json north_star: "within a stated
float tolerance"); the float32 4x4 output may differ by a few ulp.  Outlier masks and return values are exact."""
import numpy as np
import pytest

import oracle_lib
from pointslot_amd import synth

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-6


def _pose_err(Ta, Tb):
    D = np.linalg.inv(Ta.astype(np.float64)) @ Tb.astype(np.float64)
    w = np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]]) / 2
    return np.linalg.norm(w) + np.linalg.norm(D[:3, 3])


def _mat(p7):
    return oracle_lib.se3_to_mat4f(p7).astype(np.float64)


@pytest.fixture(scope="module")
def opt():
    from pointslot_amd.optimizer import Optimizer
    o = Optimizer()
    yield o
    o.close()


def test_se3_converters(opt):
    from pointslot_amd import optimizer
    rng = np.random.default_rng(2)
    for _ in range(20):
        p7 = oracle_lib.se3_exp(rng.uniform(-2, 2, 6))
        m = oracle_lib.se3_to_mat4f(p7)
        assert np.array_equal(optimizer.se3_to_mat4f(p7), m)
        assert np.allclose(optimizer.se3_from_mat4f(m), oracle_lib.se3_from_mat4f(m), atol=1e-15)


def test_pose_optimization_batch(opt):
    frames = [synth.pose_problem(0x51070003 + k) for k in range(64)]   # all 64 frames of BASELINE config 3
    frames.append(synth.pose_problem(77, n=400, mono_frac=0.5, valid_frac=0.7))
    frames.append(synth.pose_problem(78, n=14))                       # < 15 correspondences -> 0, pose untouched
    frames.append(synth.pose_problem(79, n=200, outlier_frac=0.6))
    frames.append(synth.pose_problem(80, n=3000, noise=3.0))
    frames[-1]["outlier0"] = (np.arange(3000) % 7 == 0).astype(np.uint8)
    # all information matrices zero: H = 0, lambda = 1e-5 * max|H_jj| = 0, the FIRST factorisation fails (pivot not > 0) and g2o
    # retries ten times with its zero-initialised increment (the kernel used to read uninitialised LDS there)
    frames.append(synth.pose_problem(81, n=120))
    frames[-1]["inv_sigma2"] = np.zeros_like(frames[-1]["inv_sigma2"])
    frames[65]["tcw0"] = frames[0]["tcw_true"].astype(np.float32)
    opt.enable_trace(True)
    res = opt.PoseOptimization(frames)
    noise_only = []
    for i, (f, (r, tcw, outl)) in enumerate(zip(frames, res)):
        ro, to, oo, tro = oracle_lib.pose_optimize(f, True)
        assert r == ro, (i, r, ro)
        assert np.array_equal(outl, oo), "frame %d: outlier masks differ at %d edges" % (i, (outl != oo).sum())
        assert _pose_err(tcw, to) <= POSE_TOL, (i, _pose_err(tcw, to))
        trg = opt.get_trace(i)
        if len(trg) != len(tro):
            # A round ends when a trial leaves chi2 EXACTLY unchanged (rho == 0, levenberg.cpp:151) or after ten rejections.  In an
            # iteration that has converged the difference of the two chi2 is rounding noise of sums the GPU adds in another order: one
            # side can see 0 where the other sees 1e-13 and runs one more iteration that changes nothing.  Allowed only there: every
            # iteration one side has more must leave chi2 where it was; the estimate is compared above either way.
            longer, shorter = (trg, tro) if len(trg) > len(tro) else (tro, trg)
            assert len(longer) - len(shorter) <= 2, (i, len(trg), len(tro))
            for r in longer:
                assert (np.abs(shorter[:, 0] - r[0]) <= 1e-10 * r[0]).any(), (i, r)      # no chi2 level the other run does not have
            noise_only.append(i)
            continue
        if len(tro):
            # The number of damping trials of an iteration is decided by the sign of rho = (chi2 - chi2_new) / scale.  In an
            # iteration that has already converged (its chi2 equals the previous one to ~1e-12) that difference is rounding noise
            # of the FP64 sums, which the GPU adds in a tree and g2o in edge order: there the count may differ (the estimate does
            # not: pose and outlier mask are checked above).  Everywhere else the counts must be identical.
            prev = np.concatenate([[np.inf], tro[:-1, 0]])
            converged = np.abs(prev - tro[:, 0]) <= 1e-10 * tro[:, 0]
            differ = trg[:, 2] != tro[:, 2]
            assert not (differ & ~converged).any(), (i, trg[:, 2], tro[:, 2])
            if differ.any():
                noise_only.append(i)
            assert np.allclose(trg[:, 0], tro[:, 0], rtol=1e-9)        # chi2 per iteration
            # lambda follows the gain ratio (chi - chi_new) / scale: once an iteration no longer changes chi2 that
            # ratio is pure cancellation noise, so compare lambda only where the step was significant
            prev = np.concatenate([[np.inf], tro[:-1, 0]])
            sig = (np.abs(prev - tro[:, 0]) > 1e-4 * tro[:, 0]) & (tro[:, 2] == 1)
            bad = ~np.isclose(trg[:, 1], tro[:, 1], rtol=1e-5) & sig
            assert not bad.any(), (i, trg[bad], tro[bad])
    assert np.array_equal(res[65][1], frames[65]["tcw0"])              # early return leaves the pose alone
    # (measured: 9 of the 69 frames have such an iteration; a systematic deviation would show up in many more, and in the
    # unconverged iterations, which are compared exactly above)
    assert len(noise_only) <= len(frames) // 4, noise_only
    opt.enable_trace(False)


def _cfse3_frame(seed, k, npts=120):
    rng = np.random.default_rng(seed)
    objs = []
    for o in range(k):
        p = synth.pose_problem(seed * 100 + o, n=npts + 30 * o, outlier_frac=0.15, mono_frac=0.2, valid_frac=0.8)
        # object-frame points seen through Tco = true pose; start from a perturbed pose
        T = p["tcw_true"].copy()
        Tp = T.copy(); Tp[:3, 3] += rng.uniform(-0.2, 0.2, 3)
        pose7 = oracle_lib.se3_from_mat4f(Tp.astype(np.float32))
        objs.append({"xo": p["xw"], "obs": p["obs"], "inv_sigma2": p["inv_sigma2"], "valid": p["valid"], "pose7": pose7})
    return {"objs": objs, "K": p["K"]}


def test_cfse3_batch(opt):
    frames = [_cfse3_frame(1, 1), _cfse3_frame(2, 3), _cfse3_frame(3, 6), _cfse3_frame(4, 2, npts=5), {"objs": [], "K": synth.pose_problem(1, 20)["K"]}]
    res = opt.CFSE3ObjStateOptimization(frames)
    for i, (f, (ok, poses, outls)) in enumerate(zip(frames, res)):
        oko, po, oo = oracle_lib.cfse3_optimize(f["objs"], f["K"])
        assert ok == oko, i
        for j in range(len(f["objs"])):
            assert np.array_equal(outls[j], oo[j]), (i, j)
            assert _pose_err(_mat(poses[j]), _mat(po[j])) <= POSE_TOL, (i, j)


def _ba_check(g, r, tag):
    n, po, pt, er, tr = oracle_lib.object_ba(g)
    assert len(r["trace"]) == len(tr), (tag, len(r["trace"]), len(tr))
    assert np.array_equal(r["trace"][:, 2], tr[:, 2]), (tag, r["trace"][:, 2], tr[:, 2])      # damping trials per iteration
    assert np.allclose(r["trace"][:, 0], tr[:, 0], rtol=1e-8), (tag, r["trace"][:, 0], tr[:, 0])
    assert np.array_equal(r["erase"], er), (tag, int((r["erase"] != er).sum()))
    assert r["n_erased"] == n
    for a, b in zip(r["poses"], po):
        assert _pose_err(_mat(a), _mat(b)) <= POSE_TOL, tag
    scale = max(1.0, np.abs(pt).max())
    assert np.abs(r["points"] - pt).max() <= 1e-6 * scale, tag


def test_object_ba_small_and_realistic(opt):
    graphs = [
        synth.object_ba_problem(0x51070044, n_kf=5, n_pts=14, p_vis=0.8, outlier_frac=0.0, mono_frac=0.2,
                                perturb=(0.1, 2.0, 0.05), perturb_axis="z", n_fixed_extra=1),
        synth.object_ba_problem(0x51070045, n_kf=12, n_pts=100, p_vis=0.6, perturb=(0.05, 1.0, 0.02), perturb_axis="z", n_fixed_extra=8),
        synth.object_ba_problem(0x51070046, n_kf=12, n_pts=60, p_vis=0.7, perturb_axis="z"),
        synth.object_ba_problem(0x51070047, n_kf=9, n_pts=33, p_vis=0.5, perturb_axis="y", mono_frac=0.3),
    ]
    full = synth.object_ba_problem(0x51070048, n_kf=8, n_pts=40, p_vis=0.8, noise=0.0, outlier_frac=0.0)
    full["pose_flags"] = np.where(full["pose_flags"] & 1, full["pose_flags"], 0).astype(np.uint8)   # plain SE3 vertices
    graphs.append(full)
    res = opt.ObjectLocalBundleAdjustment(graphs)
    for i, (g, r) in enumerate(zip(graphs, res)):
        _ba_check(g, r, "graph %d" % i)


def test_object_ba_config4_shape(opt):
    """BASELINE config 4 as SURVEY.md 8d defines it: 8 objects (seeds 0x51070004 + j) x 50 object keyframes x 300 points, p = 1.0
    (15 000 edges) and p = 0.6, initial poses perturbed +-0.3 m / +-5 deg yaw, points +-0.1 m - all 16 graphs in one batch, as
    bench.py runs them.  The survey's yaw perturbation turns about the camera's y axis, which VertexSE3Fix (roll / pitch locked,
    only Rz is free: src/g2o_Object.cc:190-213) cannot undo: the 5-iteration first round ends far from the optimum and the
    chi-square pass erases most observations (11 811 of 15 000 for object 0) - the reference's schedule does exactly that, and
    parity has to hold there too.  Two more graphs perturb about z (correctable) at the same magnitudes, two at a fifth of it
    (the converging regime)."""
    graphs = [synth.object_ba_problem(0x51070004 + j) for j in range(8)]
    graphs += [synth.object_ba_problem(0x51070004 + j, p_vis=0.6) for j in range(8)]
    graphs += [synth.object_ba_problem(0x51070004, perturb_axis="z"), synth.object_ba_problem(0x51070005, p_vis=0.6, perturb_axis="z")]
    graphs += [synth.object_ba_problem(0x51070004, perturb=(0.05, 1.0, 0.02), perturb_axis="z"),
               synth.object_ba_problem(0x51070005, p_vis=0.6, perturb=(0.05, 1.0, 0.02), perturb_axis="z")]
    res = opt.ObjectLocalBundleAdjustment(graphs)
    for i, (g, r) in enumerate(zip(graphs, res)):
        _ba_check(g, r, "config4 %d" % i)
    print("config-4 BA: %d graphs, %.3f ms GPU" % (len(graphs), opt.last_kernel_ms()))


def test_local_ba_shape(opt):
    """Optimizer::LocalBundleAdjustment's graph — plain SE3 keyframes (no roll/pitch lock), fixed cameras,
    many more points than poses — through the same kernels."""
    g = synth.object_ba_problem(0x51070060, n_kf=10, n_pts=1200, p_vis=0.35, perturb=(0.05, 1.0, 0.03), perturb_axis="y",
                                n_fixed_extra=6, mono_frac=0.15)
    g["pose_flags"] = (g["pose_flags"] & 1).astype(np.uint8)          # VertexSE3Expmap everywhere
    r, = opt.ObjectLocalBundleAdjustment([g])
    _ba_check(g, r, "local BA")
    assert r["n_erased"] < 0.2 * len(g["e_pose"])


def test_dynamic_static_discrimination_bit_exact():
    """Tracking::DynamicStaticDiscrimination's reprojection test: per-kind averages after the 5 x median rejection
    are bit-identical to the CPU statement (sorted, sequential FP64 sum, no FMA contraction in this kernel)."""
    from pointslot_amd.optimizer import Optimizer
    objs = [synth.dynamic_object(100 + k, n=[300, 40, 6, 1200, 2048, 5, 0, 77][k], moving=[0, 0.5, 0, 0.2, 0, 1.0, 0, 0.05][k],
                                 mono_frac=[0.3, 0.0, 0.5, 0.3, 0.5, 1.0, 0.3, 0.9][k]) for k in range(8)]
    opt = Optimizer()
    res = opt.DynamicStaticDiscrimination(objs)
    for o, r in zip(objs, res):
        e = oracle_lib.dynamic_discrimination(o)
        assert r[2] == e[2] and r[3] == e[3], (r, e)
        assert np.float64(r[0]).tobytes() == np.float64(e[0]).tobytes() and np.float64(r[1]).tobytes() == np.float64(e[1]).tobytes(), (r, e)
    assert res[1][1] > 10 * res[0][1]                       # the moving object stands out
    opt.close()


def test_randomised_optimiser_sweep():
    """pose optimisation over random sizes / outlier fractions / mono-stereo mixes / missing points and object BA over random
    graph sizes and visibilities: inlier counts, outlier masks and erase lists identical, poses within 1e-5"""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_opt.py")
    r = subprocess.run([sys.executable, tool, "7", "30"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_one_handle_shared_by_two_threads(opt):
    """The shim keeps ONE process-wide optimiser handle (host/Optimizer.h): PoseOptimization runs on the tracking thread and
    ObjectLocalBundleAdjustment on the ObjectLocalMapping thread (/root/reference/src/ObjectLocalMapping.cpp:375-377).  Calls on a
    handle are serialised inside the library: two threads hammering the same handle get exactly the single-threaded results."""
    import threading
    frames = [synth.pose_problem(0x51070003 + k) for k in range(6)]
    graphs = [synth.object_ba_problem(0x51070046, n_kf=12, n_pts=60, p_vis=0.7, perturb_axis="z"),
              synth.object_ba_problem(0x51070047, n_kf=9, n_pts=33, p_vis=0.5, mono_frac=0.3)]
    ref_pose = opt.PoseOptimization(frames)
    ref_ba = opt.ObjectLocalBundleAdjustment(graphs)
    errors = []

    def pose_loop():
        try:
            for _ in range(12):
                res = opt.PoseOptimization(frames)
                for (r, t, o), (r0, t0, o0) in zip(res, ref_pose):
                    assert r == r0 and np.array_equal(t, t0) and np.array_equal(o, o0)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    def ba_loop():
        try:
            for _ in range(6):
                res = opt.ObjectLocalBundleAdjustment(graphs)
                for a, b in zip(res, ref_ba):
                    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"]) and np.array_equal(a["erase"], b["erase"])
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=pose_loop), threading.Thread(target=ba_loop), threading.Thread(target=pose_loop)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:1]


def test_capacity_limits_are_reported_not_hidden(opt):
    """DESIGN.md section 3 lists the capacity limits of this build; each is an error code, and the handle keeps working afterwards."""
    from pointslot_amd._lib import PointslotError, PS_ERR_CAPACITY, PS_ERR_INVALID
    from pointslot_amd.extractor import ORBextractor
    big = synth.object_ba_problem(0x51070049, n_kf=130, n_pts=20, p_vis=0.3)        # 129 free poses > 128
    with pytest.raises(PointslotError) as e:
        opt.ObjectLocalBundleAdjustment([big])
    assert e.value.code == PS_ERR_CAPACITY
    ok = synth.object_ba_problem(0x51070044, n_kf=5, n_pts=14, p_vis=0.8, outlier_frac=0.0)
    assert opt.ObjectLocalBundleAdjustment([ok])[0]["iterations"] > 0                # the handle is still usable
    f = {"objs": [dict(xo=np.zeros((3, 3), np.float32), obs=np.zeros((3, 3), np.float32), inv_sigma2=np.ones(3, np.float32), valid=np.ones(3, np.uint8),
                       pose7=np.array([0, 0, 5, 0, 0, 0, 1.0]))] * 17, "K": synth.pose_problem(1, 20)["K"]}
    with pytest.raises(PointslotError) as e:                                         # 17 objects in one CFSE3 graph > 16
        opt.CFSE3ObjStateOptimization([f])
    assert e.value.code == PS_ERR_INVALID
    left, right = synth.stereo_pair()
    ex = ORBextractor(30000, 1.2, 8, 20, 5)                                          # per-level quota 6513 > 2044 (quadtree node table)
    with pytest.raises(PointslotError) as e:
        ex(left)
    assert e.value.code == PS_ERR_INVALID
    ex.close()
    ex = ORBextractor(4500, 1.2, 8, 20, 5, max_batch=2)                              # fits the extractor, but > 4096 keypoints per image in the stereo matcher
    import torch
    d = torch.from_numpy(np.stack([left, right])).cuda()
    ex.extract_batch_device(d.data_ptr(), 2, left.shape[1], left.shape[0], left.shape[1], left.size)
    with pytest.raises(PointslotError) as e:
        ex.stereo_match_batch(1, 0.53, 384.4)
    assert e.value.code == PS_ERR_CAPACITY
    kps, desc = ex.fetch(0)                                                           # extraction results are intact
    assert len(kps) > 2000
    ex.close()


def test_object_ba_error_return_leaves_the_handle_usable():
    """PS_BA_DEBUG_MAX_STEPS makes ps_object_ba_batch give up early (its only error return inside the solve loop): the call reports
    PS_ERR_HIP, frees what it created (scope guard on its events), and the same handle solves the next batch."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from pointslot_amd import synth\n"
        "from pointslot_amd._lib import PointslotError\n"
        "from pointslot_amd.optimizer import Optimizer\n"
        "import os\n"
        "opt = Optimizer()\n"
        "g = synth.object_ba_problem(0x51070004, n_kf=10, n_pts=60)\n"
        "try:\n"
        "    opt.ObjectLocalBundleAdjustment([g]); print('NOERROR')\n"
        "except PointslotError as e:\n"
        "    print('ERROR', 'did not terminate' in str(e))\n"
        "r = opt.PoseOptimization([synth.pose_problem(0x51070003)])\n"      # the same handle serves the next call
        "print('HANDLE_OK', r[0][0] > 1000)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PS_BA_DEBUG_MAX_STEPS="2")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert "ERROR True" in out.stdout and "HANDLE_OK True" in out.stdout, out.stdout + out.stderr
    # and in this process (no knob) the full schedule runs
    from pointslot_amd.optimizer import Optimizer
    opt = Optimizer()
    r, = opt.ObjectLocalBundleAdjustment([synth.object_ba_problem(0x51070004, n_kf=10, n_pts=60)])
    assert r["iterations"] > 0
    opt.close()
