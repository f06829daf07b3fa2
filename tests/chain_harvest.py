"""Test infrastructure: harvests, from the per-frame chain run over a generated drive, the inputs the hot path's per-object back end
takes in the reference - the ObjectLocalBundleAdjustment graph of one object (an ObjectKeyFrame every third frame, Tracking.cc:1475-1477
-> ObjectLocalMapping.cpp:375-377 -> Optimizer.cc:755-818) and the DynamicStaticDiscrimination problems of its frames
(Tracking.cc:2058-2202).  The chain is pointslot_amd.tracker.StereoOdometry over any backend (the HIP library on the GPU box, the CPU
checker here); the harvested numbers are what that chain produced, not synthetic graphs."""
import numpy as np

from pointslot_amd import sequence
from pointslot_amd.object_tracker import pose7, se3_from_mat4f
from pointslot_amd.tracker import StereoOdometry


def run_and_harvest(backend, seq, n_frames, kf_every=3):
    h, w = seq["left"][0].shape
    vo = StereoOdometry(backend, seq["K"], seq["bf"], w, h)
    per_obj = {}          # (track id, local-map generation) -> list of keyframes
    dyn = []
    last_tco = {}
    for k in range(n_frames):
        vo.track(seq["left"][k], seq["right"][k], sequence.frame_mask(seq, k), sequence.frame_detections(seq, k))
        ot = vo.objects
        if ot is None or ot.last is None or k < 2:
            continue
        F = ot.last
        for j, o in enumerate(F.obj):
            if o.mo is None or not o.track_ok:
                continue
            tid = F.dets[j]["id"]
            gen = id(o.mo["points"]["po"])                       # MapObjectReInit replaces the local map: ids start again
            obs = np.nonzero(o.mp_valid & o.mp_observed & (o.mp_id >= 0) & (o.outlier == 0))[0]
            tco = np.asarray(pose7(o.tco_at_frame), np.float64)
            if (tid, gen - 0) in last_tco and vo.trajectory[k] is not None and vo.trajectory[k - 1] is not None and len(obs) >= 5:
                dyn.append({"valid": np.ones(len(obs), np.uint8), "po": o.mo["points"]["po"][o.mp_id[obs]].astype(np.float64),
                            "obs": np.stack([o.x[obs], o.y[obs], o.uright[obs]], 1).astype(np.float32),
                            "inv_sigma2": ot.is2[o.octave[obs]].astype(np.float32), "last_tco": last_tco[(tid, gen)],
                            "last_tcw": np.asarray(pose7(se3_from_mat4f(vo.trajectory[k - 1])), np.float64),
                            "cur_tcw": np.asarray(pose7(se3_from_mat4f(vo.trajectory[k])), np.float64),
                            "K": tuple(np.float32(v) for v in seq["K"]), "mbf": np.float32(seq["bf"]), "frame": k, "track_id": tid})
            last_tco[(tid, gen)] = tco
            if k % kf_every == 0 and len(obs) >= 10:
                per_obj.setdefault((tid, gen), []).append({"frame": k, "tco": tco, "pid": o.mp_id[obs].copy(),
                                                           "obs": np.stack([o.x[obs], o.y[obs], o.uright[obs]], 1).astype(np.float32),
                                                           "is2": ot.is2[o.octave[obs]].astype(np.float32), "points": o.mo["points"]["po"]})
    return vo, per_obj, dyn


def ba_graph(kfs, K, bf):
    """the graph of Optimizer::ObjectLocalBundleAdjustment for one object's keyframes: the first keyframe fixed, the others VertexSE3Fix
    (pose_flags bit 1), the local-map points that at least two keyframes observe, one stereo / mono edge per observation"""
    pts_all = kfs[0]["points"]
    count = np.zeros(len(pts_all), np.int64)
    for kf in kfs:
        count[kf["pid"]] += 1
    keep = np.nonzero(count >= 2)[0]
    remap = -np.ones(len(pts_all), np.int64)
    remap[keep] = np.arange(len(keep))
    e_pose, e_point, e_obs, e_is2 = [], [], [], []
    for i, kf in enumerate(kfs):
        for p, ob, w in zip(kf["pid"], kf["obs"], kf["is2"]):
            if remap[p] >= 0:
                e_pose.append(i); e_point.append(remap[p]); e_obs.append(ob); e_is2.append(w)
    flags = np.full(len(kfs), 2, np.uint8)
    flags[0] = 3
    fx, fy, cx, cy = K
    return {"poses": np.stack([kf["tco"] for kf in kfs]).astype(np.float64), "pose_flags": flags, "points": pts_all[keep].astype(np.float64),
            "e_pose": np.array(e_pose, np.int32), "e_point": np.array(e_point, np.int32), "e_obs": np.array(e_obs, np.float32),
            "e_inv_sigma2": np.array(e_is2, np.float32),
            "K": (np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), np.float32(bf)), "frames": np.array([kf["frame"] for kf in kfs])}


def best_graph(per_obj, K, bf):
    key = max(per_obj, key=lambda k: len(per_obj[k]))
    return key[0], ba_graph(per_obj[key], K, bf)
