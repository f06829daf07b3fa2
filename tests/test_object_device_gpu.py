"""The device-resident object chain of ps_tracker (ps_tracker_step_slot_device: masks + detections in HBM, camera chain on the
background keypoints, then ExtractObjORB -> ComputeObjStereoMatches -> TrackMapObject -> SearchByBruceMatching -> CFSE3 ->
SearchByProjection(F, nOrder, MOPs) -> CFSE3 -> end of Track on the device) against its per-call twin
pointslot_amd/object_tracker.py driven (a) over the per-call C-ABI on the GPU and (b) over the CPU checker."""
import numpy as np
import pytest

from oracle_backend import OracleBackend
from pointslot_amd import sequence
from pointslot_amd.tracker import HipBackend, StereoOdometry

pytestmark = pytest.mark.gpu

INT_FIELDS = ("id", "n", "stereo", "tracked", "is_new", "track_ok", "inliers", "bf_matches", "lm_candidates", "lm_matches", "map_points")


def _run_host(backend, seq, n):
    h, w = seq["left"][0].shape
    vo = StereoOdometry(backend, seq["K"], seq["bf"], w, h)
    for k in range(n):
        vo.track(seq["left"][k], seq["right"][k], sequence.frame_mask(seq, k), sequence.frame_detections(seq, k))
    return vo


def _run_device(seqs, n, max_objects=4, max_map_objects=0):
    import torch
    from pointslot_amd.tracker_device import LockstepTracker, pack_detections
    h, w = seqs[0]["left"][0].shape
    S = len(seqs)
    trk = LockstepTracker(S, seqs[0]["K"], seqs[0]["bf"], w, h, max_steps=n, max_objects=max_objects, max_map_objects=max_map_objects)
    imgs = torch.from_numpy(np.stack([np.stack([q["left"][:n], q["right"][:n]], 1) for q in seqs], 1)).cuda()      # [n, S, 2, h, w]
    masks = torch.from_numpy(np.stack([np.stack([sequence.frame_mask(q, k) for q in seqs]) for k in range(n)])).cuda()   # [n, S, h, w]
    keep = []
    try:
        for k in range(n):
            d = torch.from_numpy(pack_detections([sequence.frame_detections(q, k) for q in seqs], max_objects).view(np.uint8)).cuda()
            keep.append(d)
            trk.step_slot_device(imgs[k].data_ptr(), masks[k].data_ptr(), d.data_ptr())
        tcw, st = trk.fetch()
        obj = trk.fetch_objects()
    finally:
        trk.close()
    return tcw, st, obj


def _check_against(vo, tcw, st, obj, s, n, exact):
    for k in range(n):
        a = vo.trajectory[k]
        if a is None:
            assert st["tracked"][k, s] == 0
        elif exact:
            assert np.array_equal(a, tcw[k, s]), "camera pose of frame %d" % k
        else:
            assert np.abs(a - tcw[k, s]).max() < 2e-5, "camera pose of frame %d" % k
        ho = vo.objects.stats[k]["objects"]
        for j, o in enumerate(ho):
            d = obj[k, s, j]
            got = {f: int(d[f]) for f in INT_FIELDS}
            want = {"id": o["id"], "n": o["n"], "stereo": o["stereo"], "tracked": int(o["tracked"]), "is_new": int(o["new"]), "track_ok": int(o["track_ok"]),
                    "inliers": o["inliers"], "bf_matches": o["bf_matches"], "lm_candidates": o["lm_candidates"], "lm_matches": o["lm_matches"],
                    "map_points": o["map_points"]}
            assert got == want, (k, j, got, want)
            # Tracking::DynamicStaticDiscrimination: the flags, the points behind the two averages, the averages themselves (bit for bit
            # against the per-call chain - same arithmetic, sums in sorted order -, to the optimiser's tolerance against the CPU checker,
            # whose object poses differ in the last digits)
            gdyn = (int(d["dynamic"]), int(d["mo_dynamic"]), int(d["dyn_n_mono"]), int(d["dyn_n_stereo"]))
            wdyn = (int(o["dynamic"]), -1 if o["mo_dynamic"] is None else int(o["mo_dynamic"]), o["dyn_n"][0], o["dyn_n"][1])
            assert gdyn == wdyn, (k, j, gdyn, wdyn)
            for f in ("dyn_mono", "dyn_stereo"):
                if exact:
                    assert float(d[f]) == o[f], (k, j, f, float(d[f]), o[f])
                else:
                    assert abs(float(d[f]) - o[f]) <= 1e-5 * max(1.0, abs(o[f])), (k, j, f, float(d[f]), o[f])
            if o["tco"] is not None:
                if exact:
                    assert np.array_equal(d["tco"], o["tco"]), (k, j, d["tco"], o["tco"])
                else:
                    assert np.abs(d["tco"] - o["tco"]).max() < 1e-6, (k, j, d["tco"], o["tco"])
        for j in range(len(ho), obj.shape[2]):
            assert obj[k, s, j]["id"] == -1


def test_device_object_chain_equals_the_per_call_chain_and_the_cpu_checker():
    n = 6
    seqs = [sequence.generate(n_frames=n, seed=4 + 3 * i, texture=sequence.kitti_texture()) for i in range(3)]
    tcw, st, obj = _run_device(seqs, n)
    for s, q in enumerate(seqs):
        be = HipBackend()
        vo = _run_host(be, q, n)
        # the per-call chain over the same kernels: integer results identical, poses bit for bit
        _check_against(vo, tcw, st, obj, s, n, exact=True)
        be.close()
    # the CPU restatement of the hot-path calls behind the same host logic: match sets identical, poses within the optimiser's tolerance
    vo = _run_host(OracleBackend(), seqs[0], n)
    _check_against(vo, tcw, st, obj, 0, n, exact=False)
    # and the objects are where the generator put them
    fx, fy, cx, cy = seqs[0]["K"]
    # (box 0 of this sequence is half hidden behind box 1: too few features, its tracking fails and MapObjectReInit runs every frame -
    # on the device exactly as in the per-call chain, which is what the comparison above covers)
    ok = 0
    for k in range(2, n):
        for b in range(2):
            if not obj[k, 0, b]["track_ok"]:
                continue
            x1, y1, x2, y2, zb, _ = seqs[0]["boxes"][k][b]
            truth = np.array([(0.5 * (x1 + x2) - cx) * zb / fx, (0.5 * (y1 + y2) - cy) * zb / fy, zb + 0.5 * sequence.BOX_DEPTH_M])
            assert np.abs(obj[k, 0, b]["tco"][:3] - truth).max() < 0.35
            ok += 1
    assert ok >= n - 2 and int(obj["reinit"][:, 0].sum()) >= 1


def test_dynamic_static_discrimination_in_the_device_chain():
    """Ten frames of a drive whose first object leaves the image-centre prior after a few frames: the reprojection test of
    Tracking::DynamicStaticDiscrimination runs inside the device chain (ob_finish) - flags, point counts and the two averages equal
    the per-call chain's (ps_dynamic_discrimination_batch behind the host logic) bit for bit and the CPU checker's."""
    n = 10
    seqs = [sequence.generate_drive(n_frames=n, seed=40, texture=sequence.kitti_texture())]
    tcw, st, obj = _run_device(seqs, n)
    be = HipBackend()
    vo = _run_host(be, seqs[0], n)
    _check_against(vo, tcw, st, obj, 0, n, exact=True)
    be.close()
    vo = _run_host(OracleBackend(), seqs[0], n)
    _check_against(vo, tcw, st, obj, 0, n, exact=False)
    ran = (obj["dyn_n_mono"][:, 0] + obj["dyn_n_stereo"][:, 0]) > 0
    assert int(ran.sum()) >= 3, "the reprojection test never ran"
    # the generator's objects move: where the test ran it says so
    assert (obj["dynamic"][:, 0][ran] == 1).all() and (obj["dyn_stereo"][:, 0][ran] > 2).all()


def test_detection_order_changes_between_frames():
    """A tracked detection whose slot differs from its slot in the last frame (ob_finish's discrimination test read the last
    frame's Tco at slot in_last while the workgroup of that slot was already replacing it): the label order is reversed on every odd frame,
    so every tracked detection has in_last != its own slot.  The device chain equals the sequential per-call chain bit for bit, three
    runs (the race was timing dependent)."""
    n = 10
    base = sequence.generate_drive(n_frames=n, seed=40, texture=sequence.kitti_texture())
    seq = dict(base)
    seq["labels"] = [list(base["labels"][k])[::-1] if k % 2 else list(base["labels"][k]) for k in range(n)]
    assert all(len(l) > 1 for l in seq["labels"])
    be = HipBackend()
    vo = _run_host(be, seq, n)
    for _ in range(3):
        tcw, st, obj = _run_device([seq], n)
        _check_against(vo, tcw, st, obj, 0, n, exact=True)
    be.close()
    _check_against(_run_host(OracleBackend(), seq, n), tcw, st, obj, 0, n, exact=False)
    ran = (obj["dyn_n_mono"][:, 0] + obj["dyn_n_stereo"][:, 0]) > 0
    assert int(ran.sum()) >= 2, "the reprojection test never ran on a detection that changed its slot"


def test_twelve_detections_per_frame_and_the_map_object_table():
    """KITTI tracking frames carry up to ~15 detections (the chain served 8): a drive with 12 objects through a tracker created
    for 16 detections per frame and 16 MapObjects per sequence equals the per-call chain bit for bit; the same drive through a tracker whose
    MapObject table holds 4 ignores the detections that would need a fifth slot and says so (PS_ERR_CAPACITY from ps_tracker_fetch_objects)."""
    n = 5
    seqs = [sequence.generate_drive(n_frames=n, seed=71, n_objects=12, texture=sequence.kitti_texture())]
    assert max(len(sequence.frame_detections(seqs[0], k)) for k in range(n)) > 8
    tcw, st, obj = _run_device(seqs, n, max_objects=16, max_map_objects=16)
    be = HipBackend()
    vo = _run_host(be, seqs[0], n)
    _check_against(vo, tcw, st, obj, 0, n, exact=True)
    be.close()
    assert int((obj["id"][n - 1, 0] >= 0).sum()) > 8 and int((obj["tracked"][n - 1, 0] != 0).sum()) >= 7
    with pytest.raises(RuntimeError, match="max_map_objects"):
        _run_device(seqs, n, max_objects=16, max_map_objects=4)


def test_device_object_chain_on_an_odd_image_size():
    """918 x 306: neither the width nor the height is a multiple of the kernels' tile / cell / row-chunk sizes (ob_masks' 256-pixel lane
    chunks, the detector's 32-pixel tiles and 8-pixel cells, the extractor's 248-column strips) - device chain = per-call chain bit for bit,
    and the CPU checker behind the same host logic agrees."""
    n = 4
    w, h = 918, 306
    K = (540.0, 540.0, 0.5 * w - 3.5, 0.5 * h + 1.5)
    seqs = [sequence.generate(n_frames=n, seed=31 + i, w=w, h=h, K=K, bf=290.0) for i in range(2)]
    tcw, st, obj = _run_device(seqs, n)
    assert int(st["tracked"][1:].sum()) == 2 * (n - 1) and int((obj["id"] >= 0).sum()) > 0
    for s, q in enumerate(seqs):
        be = HipBackend()
        vo = _run_host(be, q, n)
        _check_against(vo, tcw, st, obj, s, n, exact=True)
        be.close()
    vo = _run_host(OracleBackend(), seqs[0], n)
    _check_against(vo, tcw, st, obj, 0, n, exact=False)


def test_forward_drive_with_yaw_device_chain_host_chain_and_cpu_checker():
    """The KITTI-like scene (sequence.generate_drive: forward motion with yaw, exact ray-cast planes): SearchByProjection(cur, last)
    takes its forward branch, keypoints change octave, local-map points leave the frustum.  Device chain = per-call chain bit for
    bit, = the CPU checker's match sets with poses within the optimiser's tolerance; the trajectory follows the ground truth."""
    n = 7
    seqs = [sequence.generate_drive(n_frames=n, seed=60 + i, speed=0.6 + 0.1 * i, yaw_rate_deg=0.4 + 0.2 * i, texture=sequence.kitti_texture()) for i in range(2)]
    tcw, st, obj = _run_device(seqs, n)
    for s, q in enumerate(seqs):
        be = HipBackend()
        vo = _run_host(be, q, n)
        _check_against(vo, tcw, st, obj, s, n, exact=True)
        be.close()
        for k in range(n):
            assert st["tracked"][k, s] == 1
            twc = -(tcw[k, s, :3, :3].T @ tcw[k, s, :3, 3])
            assert np.abs(twc - q["twc"][k][:, 3]).max() < 0.06, (s, k, twc, q["twc"][k][:, 3])
        assert twc[2] > 2.5                                   # the camera really drove forward
    vo = _run_host(OracleBackend(), seqs[0], n)
    _check_against(vo, tcw, st, obj, 0, n, exact=False)
    assert int((obj["track_ok"][2:, 0, 0] != 0).sum()) == n - 2        # the vehicle ahead; the far one at the side has too few features at first


def test_overflow_is_reported_per_sequence_and_step_not_by_a_failing_fetch():
    import torch
    from pointslot_amd._lib import lib, check
    from pointslot_amd.tracker_device import LockstepTracker
    seqs = [sequence.generate(n_frames=3, seed=70 + i) for i in range(3)]
    h, w = seqs[0]["left"][0].shape
    trk = LockstepTracker(3, seqs[0]["K"], seqs[0]["bf"], w, h, max_steps=3)
    imgs = torch.from_numpy(np.stack([np.stack([q["left"], q["right"]], 1) for q in seqs], 1)).cuda()
    trk.step_device(imgs[0].data_ptr())
    check(lib.ps_tracker_debug_set_overflow(trk._h, 1, 5))     # as if five windows of sequence 1 overflowed in the next step
    trk.step_device(imgs[1].data_ptr())
    trk.step_device(imgs[2].data_ptr())
    tcw, st = trk.fetch()                                       # succeeds: the other sequences' results are not withheld
    assert st["overflowed"].tolist() == [[0, 0, 0], [0, 5, 0], [0, 0, 0]]
    assert (st["tracked"] == 1).all()
    trk.close()


def test_a_detector_overflow_in_an_earlier_queued_step_is_not_lost():
    """The batched cv::ORB clears its overflow flags at the start of every step; the tracker is meant to queue many steps before a fetch.
    A frame of noise under an all-object mask overflows the per-level candidate capacity in step 0 of sequence 1; two ordinary steps
    follow; ps_tracker_fetch_objects must still say so (per-sequence sticky counters, cleared by ps_tracker_reset)."""
    import torch
    from pointslot_amd._lib import PointslotError
    from pointslot_amd.tracker_device import LockstepTracker, pack_detections
    n, S = 3, 2
    seqs = [sequence.generate(n_frames=n, seed=90 + i) for i in range(S)]
    h, w = seqs[0]["left"][0].shape
    rng = np.random.default_rng(5)
    imgs = np.stack([np.stack([q["left"][:n], q["right"][:n]], 1) for q in seqs], 1).copy()      # [n, S, 2, h, w]
    masks = np.stack([np.stack([sequence.frame_mask(q, k) for q in seqs]) for k in range(n)]).copy()
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    imgs[0, 1, 0] = noise; imgs[0, 1, 1] = noise
    masks[0, 1] = 1                                                    # every pixel belongs to detection id 0
    d_imgs, d_masks = torch.from_numpy(imgs).cuda(), torch.from_numpy(masks).cuda()
    trk = LockstepTracker(S, seqs[0]["K"], seqs[0]["bf"], w, h, max_steps=2 * n, max_objects=4)
    keep = []

    def run():
        for k in range(n):
            d = torch.from_numpy(pack_detections([sequence.frame_detections(q, k) for q in seqs], 4).view(np.uint8)).cuda()
            keep.append(d)
            trk.step_slot_device(d_imgs[k].data_ptr(), d_masks[k].data_ptr(), d.data_ptr())
    run()
    with pytest.raises(PointslotError) as e:
        trk.fetch_objects()
    assert "sequence 1" in str(e.value) and "capacity" in str(e.value)
    tcw, st = trk.fetch()                                              # the camera results are not withheld
    assert tcw.shape[0] == n
    # after a reset the counters are clear: the same tracker on ordinary frames only
    trk.reset()
    imgs[0, 1, 0] = seqs[1]["left"][0]; imgs[0, 1, 1] = seqs[1]["right"][0]
    masks[0, 1] = sequence.frame_mask(seqs[1], 0)
    d_imgs, d_masks = torch.from_numpy(imgs).cuda(), torch.from_numpy(masks).cuda()
    run()
    obj = trk.fetch_objects()
    assert (obj["id"][:, :, :2] >= 0).all()
    trk.close()


def test_object_features_on_a_second_stream_give_the_same_results():
    """PS_TRK_OVERLAP=1 (ExtractObjORB on a second stream beside the camera chain, joined before ComputeObjStereoMatches) is a
    scheduling option: every pose, statistic and object record must come out bit for bit as on one stream.  The option is read when
    the handle is created, so the second run happens in a fresh process."""
    import os
    import pickle
    import subprocess
    import sys
    n = 5
    code = ("import sys, pickle, numpy as np\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from pointslot_amd import sequence\n"
            "from test_object_device_gpu import _run_device\n"
            "seqs = [sequence.generate_drive(n_frames=%d, seed=81 + i, texture=sequence.kitti_texture()) for i in range(3)]\n"
            "tcw, st, obj = _run_device(seqs, %d)\n"
            "sys.stdout.buffer.write(pickle.dumps((tcw.tobytes(), st.tobytes(), obj.tobytes())))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), n, n))
    outs = []
    for overlap in (False, True):
        env = dict(os.environ)
        env.pop("PS_TRK_OVERLAP", None)
        if overlap:
            env["PS_TRK_OVERLAP"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(pickle.loads(r.stdout))
    assert outs[0] == outs[1]
    assert len(outs[0][2]) > 0


def test_object_head_on_the_second_stream_changes_nothing(monkeypatch):
    """r06: a tracker of a few sequences runs masks / cv::ORB / ComputeObjStereoMatches / AssignFeatures on a second stream beside the camera
    chain (PS_TRK_OVERLAP, default on up to 32 sequences).  With it forced off and on: every pose, every camera statistic and every object
    record bit for bit the same."""
    n = 6
    seqs = [sequence.generate_drive(n_frames=n, seed=40 + 7 * i, texture=sequence.kitti_texture()) for i in range(2)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PS_TRK_OVERLAP", mode)
        res[mode] = _run_device(seqs, n)
    for a, b in zip(res["0"], res["1"]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert int((res["1"][2]["track_ok"] != 0).sum()) > n
