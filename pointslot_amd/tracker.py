"""The caller either side of the hot path: a stereo visual-odometry loop that chains the kernels the way the reference's
tracking thread does in localisation mode (`mbOnlyTracking`), so that the path can be exercised and measured end to end on a
sequence:

    Frame::Frame            ExtractORB x2 (a2-a8)  ->  ComputeStereoMatches (8f-1)           src/Frame.cc:709-722
    StereoInitialization    map points from every keypoint with depth                        src/Tracking.cc:2840-2910
    UpdateLastFrame         temporal "visual odometry" points, closest first                 src/Tracking.cc:2971-3026
    TrackWithMotionModel    SearchByProjection(cur, last, th=7) (a11) -> PoseOptimization (a14)   src/Tracking.cc:3028-3095
    TrackLocalMap           isInFrustum + SearchByProjection(F, points, th) (a12) -> PoseOptimization   src/Tracking.cc:3097-3160
    motion model            mVelocity = Tcw * LastTwc                                        src/Tracking.cc:1260-1272

With `mask` and `detections` (SLOT.MODE 4) `track()` also runs the object half of Tracking::Track on the frame - AssignFeatures on
the static keypoints here, and `object_tracker.ObjectTracker` for ExtractObjORB, ComputeObjStereoMatches, TrackMapObject,
TrackLastFrameObjectPoint and TrackObjectLocalMap (src/Tracking.cc:1224-1233, 2288-2712) over four more backend calls.

Only this data flow is kept; keyframes, local mapping, relocalisation and loop closing are the reference's control plane and are
not rebuilt.  The loop is backend-agnostic: the product backend below drives libpointslot_hip.so; the tests plug the CPU checker in
through the same calls and compare trajectories and object records.  Matrix products and inverses are written out operation by
operation in float32 (`mul4`, `inverse_rt`), as the glue kernels of the device-resident chain evaluate them, so that the two
chains agree bit for bit.
"""
import numpy as np

from .matcher import build_grid, FRAME_GRID_COLS, FRAME_GRID_ROWS


class HipBackend:
    """The hot-path calls on the GPU: ORBextractor x2, ComputeStereoMatches, two SearchByProjection overloads, PoseOptimization;
    for the object half cv::ORB x2 + ComputeObjStereoMatches, SearchByBruceMatching, SearchByProjection(F, nOrder, MOPs), CFSE3."""

    def __init__(self, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=5, device=0):
        from .extractor import ORBextractor
        from .matcher import ORBmatcher
        from .optimizer import Optimizer
        self.left = ORBextractor(nfeatures, scale, nlevels, ini_th, min_th, device=device)
        self.right = ORBextractor(nfeatures, scale, nlevels, ini_th, min_th, device=device)
        self.matcher_mm = ORBmatcher(0.9, True, device=device)     # Tracking.cc:3030
        self.matcher_lm = ORBmatcher(0.8, True, device=device)     # Tracking.cc SearchLocalPoints
        self.optimizer = Optimizer(device=device)
        self.device = device
        self.cv_left = self.cv_right = None                        # the object detector pair, created on first use
        self.scale_factors = self.left.GetScaleFactors()
        self.inv_level_sigma2 = self.left.GetInverseScaleSigmaSquares()

    def extract_stereo(self, left, right, mb, mbf):
        from .extractor import ComputeStereoMatches
        kps, desc = self.left(left)
        self.right(right)
        ur, dp = ComputeStereoMatches(self.left, self.right, mb, mbf)
        return kps, desc, ur, dp

    def search_frame(self, problem):
        return self.matcher_mm.SearchByProjection([problem])[0]

    def search_points(self, problem):
        return self.matcher_lm.SearchByProjection([problem])[0]

    def pose_optimization(self, frame):
        return self.optimizer.PoseOptimization([frame])[0]

    # ---- the object half (object_tracker.ObjectTracker) ----
    def extract_objects(self, left, right, mask_left, mask_right, mb, mbf):
        """Frame::ExtractObjORB + ComputeObjStereoMatches: cv::ORB(1000, 1.2, 8, 19) under the object masks, then the stereo matcher
        on those key sets against the pyramids the two ORBextractors hold for the SAME frame (extract_stereo ran before)."""
        from .extractor import ComputeObjStereoMatches
        if self.cv_left is None:
            from .object_orb import ORB
            self.cv_left, self.cv_right = ORB(1000, 1.2, 8, 19, device=self.device), ORB(1000, 1.2, 8, 19, device=self.device)
        kl, dl = self.cv_left.detectAndCompute(left, mask_left)
        kr, dr = self.cv_right.detectAndCompute(right, mask_right)
        if len(kl) == 0:
            return kl, dl, np.zeros(0, np.float32), np.zeros(0, np.float32)
        ur, dp, _ = ComputeObjStereoMatches(self.left, self.right, kl, dl, kr, dr, mb, mbf)
        return kl, dl, ur, dp

    def search_bruteforce(self, problems):
        return self.matcher_mm.SearchByBruceMatching(problems)          # ORBmatcher matcher(0.9, true), Tracking.cc:2381

    def search_object_points(self, problems):
        return self.matcher_lm.SearchByProjection(problems)             # ORBmatcher matcher(0.8), Tracking.cc:2569

    def cfse3(self, objs, K):
        return self.optimizer.CFSE3ObjStateOptimization([{"objs": objs, "K": K}])[0]

    def dynamic_discrimination(self, objs):
        return self.optimizer.DynamicStaticDiscrimination(objs)        # the reprojection test of Tracking::DynamicStaticDiscrimination

    def close(self):
        for o in (self.left, self.right, self.matcher_mm, self.matcher_lm, self.optimizer, self.cv_left, self.cv_right):
            if o is not None:
                o.close()


class _Frame:
    pass


def _f32(x):
    return np.asarray(x, np.float32)


# float32 4x4 products written out as the device kernels (track_kernels.hip: mul4) and the C++ host class evaluate them - sum over k
# in order, every product and sum rounded to float32 - so that the three drivers agree bit for bit (a BLAS matmul may fuse or reorder)
def mul4(a, b):
    a = _f32(a); b = _f32(b)
    out = np.zeros((4, 4), np.float32)
    for r in range(4):
        for c in range(4):
            acc = np.float32(0)
            for k in range(4):
                acc = np.float32(acc + np.float32(a[r, k] * b[k, c]))
            out[r, c] = acc
    return out


def inverse_rt(t):
    """[R | t]^-1 = [R^T | -(R^T t)] as Tracking.cc:1260-1268 builds LastTwc (the sum over the rows in order)"""
    t = _f32(t)
    out = np.eye(4, dtype=np.float32)
    for r in range(3):
        for c in range(3):
            out[r, c] = t[c, r]
    for r in range(3):
        acc = np.float32(0)
        for c in range(3):
            acc = np.float32(acc + np.float32(t[c, r] * t[c, 3]))
        out[r, 3] = -acc
    return out


class StereoOdometry:
    """State of the tracking thread that the hot path needs: last frame, motion model, the initial keyframe's map points."""

    def __init__(self, backend, K, bf, width, height, th_depth=35.0, track_local_map=True):
        self.be = backend
        self.fx, self.fy, self.cx, self.cy = [np.float32(v) for v in K]
        self.bf = np.float32(bf)
        self.mb = np.float32(self.bf / self.fx)                         # Frame.cc: mb = mbf / fx
        self.th_depth = np.float32(self.bf * np.float32(th_depth) / self.fx)   # Tracking.cc:402
        self.w, self.h = width, height
        self.grid = (np.float32(0), np.float32(0), np.float32(FRAME_GRID_COLS) / np.float32(width),
                     np.float32(FRAME_GRID_ROWS) / np.float32(height))    # Frame.cc:1636-1640 without distortion
        self.sf = _f32(backend.scale_factors)
        self.log_sf = np.float32(np.log(self.sf[1]))
        self.last = None
        self.velocity = None
        self.local_map = None
        self.track_local_map = track_local_map
        self.state = "NOT_INITIALIZED"
        self.prev_tcw = None            # mLastFrame.mTcw (None: the frame before had no pose)
        self.objects = None             # object_tracker.ObjectTracker once a frame came with a mask
        self.trajectory = []            # Tcw per frame (float32 4x4), None when lost
        self.stats = []

    # ---- Frame::Frame (stereo) ----
    def _make_frame(self, left, right, mask=None):
        F = _Frame()
        F.kps, F.desc, F.u_right, F.depth = self.be.extract_stereo(left, right, self.mb, self.bf)
        if mask is not None and len(F.kps):
            # Frame::AssignFeatures (Frame.cc:762-977, after ComputeStereoMatches): only keypoints on background pixels
            # (mask 0) stay static features; object / ignored pixels leave the static set
            keep = mask[F.kps["y"].astype(np.int64), F.kps["x"].astype(np.int64)] == 0
            F.kps, F.desc, F.u_right, F.depth = F.kps[keep], F.desc[keep], F.u_right[keep], F.depth[keep]
        F.N = len(F.kps)
        F.x, F.y = _f32(F.kps["x"]), _f32(F.kps["y"])
        F.octave = np.asarray(F.kps["octave"], np.int32)
        F.angle = _f32(F.kps["angle"])
        F.cell_off, F.cell_idx = build_grid(F.x, F.y, *self.grid)
        F.mp_xw = np.zeros((F.N, 3), np.float32)      # mvpMapPoints[i]->GetWorldPos()
        F.mp_valid = np.zeros(F.N, bool)              # mvpMapPoints[i] != NULL
        F.mp_observed = np.zeros(F.N, bool)           # Observations() > 0 (false for temporal points)
        F.mp_id = np.full(F.N, -1, np.int64)          # index into the local map (the initial keyframe's points)
        F.outlier = np.zeros(F.N, np.uint8)
        F.tcw = None
        return F

    def _unproject(self, F, idx):
        """Frame::UnprojectStereo (Frame.cc:2505-2519), float arithmetic"""
        z = F.depth[idx]
        x = (F.x[idx] - self.cx) * z * (np.float32(1) / self.fx)
        y = (F.y[idx] - self.cy) * z * (np.float32(1) / self.fy)
        T = F.tcw
        out = np.zeros((len(x), 3), np.float32)
        for r in range(3):
            ow = -((T[0, r] * T[0, 3] + T[1, r] * T[1, 3]) + T[2, r] * T[2, 3])          # mOw = -Rcw^T tcw
            out[:, r] = ((T[0, r] * x + T[1, r] * y) + T[2, r] * z) + ow
        return out

    def _train(self, F):
        return {"x": F.x, "y": F.y, "octave": F.octave, "angle": F.angle, "u_right": F.u_right, "desc": F.desc,
                "occupied": (F.mp_valid & F.mp_observed).astype(np.uint8), "grid": self.grid,
                "cell_off": F.cell_off, "cell_idx": F.cell_idx}

    def _K6(self):
        return (self.fx, self.fy, self.cx, self.cy, self.bf, self.mb)

    # ---- Tracking::StereoInitialization ----
    def _initialize(self, F):
        if F.N <= 500:
            return False
        F.tcw = np.eye(4, dtype=np.float32)
        idx = np.nonzero(F.depth > 0)[0]
        F.mp_xw[idx] = self._unproject(F, idx)
        F.mp_valid[idx] = True
        F.mp_observed[idx] = True
        F.mp_id[idx] = np.arange(len(idx))
        # MapPoint::UpdateNormalAndDepth for one observation (MapPoint.cc:470-497)
        PO = F.mp_xw[idx]                              # camera centre of the initial keyframe is the origin
        dist = np.sqrt((PO[:, 0] * PO[:, 0] + PO[:, 1] * PO[:, 1]) + PO[:, 2] * PO[:, 2]).astype(np.float32)
        maxd = (dist * self.sf[F.octave[idx]]).astype(np.float32)
        self.local_map = {"xw": F.mp_xw[idx].copy(), "desc": F.desc[idx].copy(), "normal": (PO / dist[:, None]).astype(np.float32),
                          "max_dist": maxd, "min_dist": (maxd / self.sf[-1]).astype(np.float32)}
        self.state = "OK"
        return True

    # ---- Tracking::UpdateLastFrame (localisation mode) ----
    def _update_last_frame(self):
        L = self.last
        idx = np.nonzero(L.depth > 0)[0]
        if len(idx) == 0:
            return
        order = idx[np.lexsort((idx, L.depth[idx]))]           # sort(pair<float,int>)
        create = []
        npoints = 0
        for i in order:
            if not L.mp_valid[i] or not L.mp_observed[i]:
                create.append(i)
            npoints += 1
            if L.depth[i] > 2 * self.th_depth and npoints > 100:
                break
        if create:
            create = np.asarray(create)
            L.mp_xw[create] = self._unproject(L, create)
            L.mp_valid[create] = True
            L.mp_observed[create] = False
            L.mp_id[create] = -1

    # ---- Tracking::TrackWithMotionModel ----
    def _track_motion_model(self, F):
        L = self.last
        self._update_last_frame()
        F.tcw = mul4(self.velocity, L.tcw)
        query = {"valid": (L.mp_valid & (L.outlier == 0)).astype(np.uint8), "desc": L.desc,
                 "observed": np.ones(L.N, np.uint8), "angle": L.angle, "xw": L.mp_xw, "octave": L.octave}
        nm = 0
        for th in (7.0, 14.0):                                  # th = 7 (stereo), then 2 * th
            F.mp_valid[:] = False
            pr = {"mode": "frame", "train": self._train(F), "query": query, "tcw": F.tcw, "tlw": L.tcw, "K6": self._K6(),
                  "bounds": (0.0, float(self.w), 0.0, float(self.h)), "scale_factors": self.sf, "th": th, "mono": False}
            nm, match = self.be.search_frame(pr)
            if nm >= 20:
                break
        if nm < 20:
            return False, nm, 0
        m = match >= 0
        F.mp_valid[:] = m
        F.mp_xw[m] = L.mp_xw[match[m]]
        F.mp_observed[m] = L.mp_observed[match[m]]
        F.mp_id[m] = L.mp_id[match[m]]
        self._pose_optimization(F)
        # discard outliers (Tracking.cc:3062-3082)
        out = F.mp_valid & (F.outlier != 0)
        F.mp_valid[out] = False
        F.outlier[out] = 0
        nmatches = int(F.mp_valid.sum())
        nmatches_map = int((F.mp_valid & F.mp_observed).sum())
        self.vo = nmatches_map < 10
        return nmatches > 20, nmatches, nmatches_map

    def _pose_optimization(self, F):
        obs = np.stack([F.x, F.y, F.u_right], 1).astype(np.float32)
        frame = {"xw": F.mp_xw, "obs": obs, "inv_sigma2": _f32(self.be.inv_level_sigma2)[F.octave], "valid": F.mp_valid.astype(np.uint8),
                 "K": (self.fx, self.fy, self.cx, self.cy, self.bf), "tcw0": F.tcw, "outlier0": F.outlier}
        ninl, tcw, outlier = self.be.pose_optimization(frame)
        F.outlier = np.asarray(outlier, np.uint8).copy()
        if int(F.mp_valid.sum()) >= 15:       # Optimizer.cc:376-377: fewer than 15 edges -> return 0 before SetPose
            F.tcw = _f32(tcw).copy()
        return ninl

    # ---- Tracking::SearchLocalPoints + TrackLocalMap ----
    def _track_local_map(self, F):
        M = self.local_map
        n = len(M["xw"])
        already = np.zeros(n, bool)
        # mnLastFrameSeen == mCurrentFrame.mnId: the frame's map points, and the ones PoseOptimization just discarded as outliers
        # (Tracking.cc:3071-3075 stamps them too; their slot keeps the id)
        ids = F.mp_id[F.mp_id >= 0]
        already[ids] = True
        # Frame::isInFrustum (Frame.cc:1686-1743), float arithmetic, viewingCosLimit 0.5
        T = F.tcw
        X, Y, Z = M["xw"][:, 0], M["xw"][:, 1], M["xw"][:, 2]
        Pc = [((T[r, 0] * X + T[r, 1] * Y) + T[r, 2] * Z) + T[r, 3] for r in range(3)]
        z = Pc[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            invz = (np.float32(1) / z).astype(np.float32)
            u = (self.fx * Pc[0] * invz + self.cx).astype(np.float32)
            v = (self.fy * Pc[1] * invz + self.cy).astype(np.float32)
        Ow = [-((T[0, r] * T[0, 3] + T[1, r] * T[1, 3]) + T[2, r] * T[2, 3]) for r in range(3)]
        PO = np.stack([X - Ow[0], Y - Ow[1], Z - Ow[2]], 1).astype(np.float32)
        dist = np.sqrt((PO[:, 0] * PO[:, 0] + PO[:, 1] * PO[:, 1]) + PO[:, 2] * PO[:, 2]).astype(np.float32)
        N = M["normal"]
        with np.errstate(divide="ignore", invalid="ignore"):
            view_cos = (((PO[:, 0] * N[:, 0] + PO[:, 1] * N[:, 1]) + PO[:, 2] * N[:, 2]) / dist).astype(np.float32)
            ratio = (M["max_dist"] / dist).astype(np.float32)
            # MapPoint::PredictScale: the float logarithm taken as the rounded double one (as the device kernel does)
            level = np.ceil(np.log(ratio.astype(np.float64)).astype(np.float32) / self.log_sf)
        ok = ~already & ~(z < 0) & ~(u < 0) & ~(u > self.w) & ~(v < 0) & ~(v > self.h)
        ok &= ~(dist < np.float32(0.8) * M["min_dist"]) & ~(dist > np.float32(1.2) * M["max_dist"]) & ~(view_cos < np.float32(0.5))
        level = np.clip(np.nan_to_num(level, nan=0.0, posinf=7.0, neginf=0.0), 0, len(self.sf) - 1).astype(np.int32)
        nto = int(ok.sum())
        nfound = 0
        if nto > 0:
            query = {"valid": ok.astype(np.uint8), "desc": M["desc"], "observed": np.ones(n, np.uint8), "proj_x": np.where(ok, u, 0).astype(np.float32),
                     "proj_y": np.where(ok, v, 0).astype(np.float32), "proj_xr": np.where(ok, u - self.bf * invz, 0).astype(np.float32),
                     "level": level, "view_cos": np.where(ok, view_cos, 0).astype(np.float32)}
            pr = {"mode": "points", "train": self._train(F), "query": query, "scale_factors": self.sf, "th": 1.0}
            nfound, match = self.be.search_points(pr)
            m = match >= 0                 # occupied slots are never matched; temporal matches are overwritten (ORBmatcher.cc:146)
            F.mp_valid[m] = True
            F.mp_xw[m] = M["xw"][match[m]]
            F.mp_observed[m] = True
            F.mp_id[m] = match[m]
        self._pose_optimization(F)
        inl = F.mp_valid & (F.outlier == 0)
        # stereo: outliers lose their map point (Tracking.cc:3141-3142)
        drop = F.mp_valid & (F.outlier != 0)
        F.mp_valid[drop] = False
        return int(inl.sum()) >= 30, int(inl.sum()), nfound

    # ---- Tracking::Track for one stereo frame ----
    def track(self, left, right, mask=None, detections=None):
        """mask / detections (SLOT.MODE 4): the frame's 8-bit instance-id mask (Frame::ReadKittiSegmentationImage: 0 background,
        255 ignored, id + 1 on object pixels) and its offline detections (object_tracker.detection_from_label); with them the
        static features are the background keypoints and the object chain runs after the camera chain, as in Tracking::Track."""
        F = self._make_frame(left, right, mask)
        was_initialized = self.state != "NOT_INITIALIZED"
        prev_tcw = self.prev_tcw
        tcw = self._track_camera(F)
        self.prev_tcw = None if tcw is None else tcw.copy()
        if mask is not None:
            if self.objects is None:
                from .object_tracker import ObjectTracker
                self.objects = ObjectTracker(self.be, (self.fx, self.fy, self.cx, self.cy), self.bf, self.w, self.h, self.th_depth, self.grid,
                                             self.sf, self.be.inv_level_sigma2)
            self.objects.track(left, right, mask, detections or [], tcw, prev_tcw, was_initialized)
        return tcw

    def _track_camera(self, F):
        st = {"N": F.N, "stereo": int((F.depth > 0).sum())}
        if self.state == "NOT_INITIALIZED":
            if self._initialize(F):
                self.last = F
                self.velocity = None
            self.trajectory.append(None if F.tcw is None else F.tcw.copy())
            self.stats.append(st)
            return F.tcw
        if self.velocity is None:
            # the reference takes TrackReferenceKeyFrame (BoW matching against ORBvoc) for the frame right after
            # initialisation (Tracking.cc:1141-1145); without the vocabulary blob this slice substitutes the motion model
            # with an identity velocity, as SURVEY.md 8d config 1 prescribes
            self.velocity = np.eye(4, dtype=np.float32)
        ok, nm, nmap = self._track_motion_model(F)
        st.update(matches=nm, map_matches=nmap)
        if ok and self.track_local_map and not self.vo:
            ok, ninl, nfound = self._track_local_map(F)
            st.update(local_inliers=ninl, local_found=nfound)
        if not ok:
            self.state = "LOST"
            self.trajectory.append(None)
            self.stats.append(st)
            self.velocity = None
            return None
        # motion model (Tracking.cc:1260-1270)
        L = self.last
        self.velocity = mul4(F.tcw, inverse_rt(L.tcw))
        # clean VO matches (Tracking.cc:1274-1286): matches to temporal points do not outlive the frame
        tmp = F.mp_valid & ~F.mp_observed
        F.mp_valid[tmp] = False
        F.outlier[tmp] = 0
        self.last = F
        self.state = "OK"                # `if (bOK) mState = OK;` - also after a frame that was lost
        self.trajectory.append(F.tcw.copy())
        self.stats.append(st)
        return F.tcw


def save_trajectory_kitti(path, trajectory):
    """System::SaveTrajectoryKITTI line format (src/System.cc:395-402): row-major 3x4 [Rwc | twc], setprecision(9)."""
    with open(path, "w") as f:
        for tcw in trajectory:
            if tcw is None:
                continue
            Rwc = tcw[:3, :3].T
            twc = -(Rwc @ tcw[:3, 3])
            M = np.concatenate([Rwc, twc[:, None]], 1).astype(np.float32)
            f.write(" ".join("%.9g" % float(v) for v in M.reshape(12)) + "\n")
