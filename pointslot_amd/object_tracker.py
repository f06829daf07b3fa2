"""The object half of the tracking thread's per-frame chain (SLOT.MODE 4: offline detections + instance masks), driven call
by call through the hot-path kernels - the per-call twin of the device-resident object chain of ps_tracker:

    Frame::Frame            ExtractObjORB: cv::ORB(1000, 1.2, 8, 19) on left / right under the object masks (8f-2)
                            ComputeObjStereoMatches (8f-1), AssignFeatures                    src/Frame.cc:690-733,762-977,2623-2665
    TrackMapObject          pose prediction Tcl * Tco, InitializeCurrentObjPose (RANSAC centroid), FineTuningUsing2dBox,
                            MapObjectInit for a first observation                             src/Tracking.cc:1533-1930
    TrackLastFrameObjectPoint   temporal points, SearchByBruceMatching (a10), CFSE3ObjStateOptimization (a15)   :2288-2466
    TrackObjectLocalMap     isInFrustum(pMP, nOrder), SearchByProjection(F, nOrder, MOPs) (a13), CFSE3 (a15)    :2468-2712
    DynamicStaticDiscrimination   depth / image-centre gates, reprojection test "the object did not move" (8f-4),
                            DetectionObject::SetDynamicFlag, MapObject::DynamicDetection / SetDynamicFlag           :1244,2058-2202
    end of Track            temporal matches dropped, MapObjectReInit for an object whose tracking failed       :1443-1478,1932-2031

The slice is the localisation-mode one of tracker.py: an object's local map is the object keyframe of its (re-)initialisation
(no NeedNewObjectKeyFrame / ObjectLocalMapping); DynamicStaticDiscrimination runs where Track runs it (after TrackObjectLocalMap,
r05) and keeps the detections' and the MapObjects' dynamic flags, its tail StaticPointRecoveryFromObj (it moves object points into
the static map) is outside the slice, and an object's virtual velocity stays zero (MapObject::UpdateVelocity is outside the slice;
the prediction is Tcl * Tco as the reference has it for an object without a velocity).

Arithmetic is written operation by operation in the precision the reference uses (float32 pixel / point arithmetic, FP64 pose
algebra on g2o::SE3Quat) so that the device-resident chain can repeat it bit for bit.  Where OpenCV's cv::Mat arithmetic decides a
rounding (cv::norm, Mat / double, the float gemm behind -Roc * tco) the statement here is: norms accumulate in double and round
once, products and sums of float matrices stay in float - modelling choices like the others of DESIGN.md section 2.
"""
import math

import numpy as np

from .matcher import build_grid

F32 = np.float32


# ---- g2o::SE3Quat on Python doubles: the operation order of pointslot_amd/csrc/se3.h (no contraction, IEEE sqrt / divide) ----
def q_rotate(q, v):
    ux = q[1] * v[2] - q[2] * v[1]; uy = q[2] * v[0] - q[0] * v[2]; uz = q[0] * v[1] - q[1] * v[0]
    ux += ux; uy += uy; uz += uz
    return (v[0] + q[3] * ux + (q[1] * uz - q[2] * uy),
            v[1] + q[3] * uy + (q[2] * ux - q[0] * uz),
            v[2] + q[3] * uz + (q[0] * uy - q[1] * ux))


def se3_normalize(t, q):
    q = list(q)
    if q[3] < 0:
        q = [-q[0], -q[1], -q[2], -q[3]]
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    return (tuple(t), (q[0] / n, q[1] / n, q[2] / n, q[3] / n))


def se3_mul(a, b):
    """SE3Quat::operator* (se3quat.h:104-110); a, b = (t, q)"""
    (at, aq), (bt, bq) = a, b
    rt = q_rotate(aq, bt)
    t = (at[0] + rt[0], at[1] + rt[1], at[2] + rt[2])
    w = aq[3] * bq[3] - aq[0] * bq[0] - aq[1] * bq[1] - aq[2] * bq[2]
    x = aq[3] * bq[0] + aq[0] * bq[3] + aq[1] * bq[2] - aq[2] * bq[1]
    y = aq[3] * bq[1] + aq[1] * bq[3] + aq[2] * bq[0] - aq[0] * bq[2]
    z = aq[3] * bq[2] + aq[2] * bq[3] + aq[0] * bq[1] - aq[1] * bq[0]
    return se3_normalize(t, (x, y, z, w))


def se3_inverse(a):
    """SE3Quat::inverse (se3quat.h:112-117): conjugate, t = q^-1 * (t * -1)"""
    t, q = a
    qi = (-q[0], -q[1], -q[2], q[3])
    return (q_rotate(qi, (t[0] * -1.0, t[1] * -1.0, t[2] * -1.0)), qi)


def se3_map(a, x):
    r = q_rotate(a[1], x)
    return (r[0] + a[0][0], r[1] + a[0][1], r[2] + a[0][2])


def quat_to_R(q):
    tx = 2 * q[0]; ty = 2 * q[1]; tz = 2 * q[2]
    twx = tx * q[3]; twy = ty * q[3]; twz = tz * q[3]
    txx = tx * q[0]; txy = ty * q[0]; txz = tz * q[0]
    tyy = ty * q[1]; tyz = tz * q[1]; tzz = tz * q[2]
    return ((1 - (tyy + tzz), txy - twz, txz + twy), (txy + twz, 1 - (txx + tzz), tyz - twx), (txz - twy, tyz + twx, 1 - (txx + tyy)))


def quat_from_R(R):
    """Eigen::Quaterniond(Matrix3d) (se3.h: se3_quat_from_R); R row-major 3x3 of doubles"""
    tr = R[0][0] + R[1][1] + R[2][2]
    if tr > 0:
        s = math.sqrt(tr + 1.0); w = 0.5 * s; s = 0.5 / s
        return ((R[2][1] - R[1][2]) * s, (R[0][2] - R[2][0]) * s, (R[1][0] - R[0][1]) * s, w)
    if R[0][0] >= R[1][1] and R[0][0] >= R[2][2]:
        s = math.sqrt(R[0][0] - R[1][1] - R[2][2] + 1.0); x = 0.5 * s; s = 0.5 / s
        return (x, (R[1][0] + R[0][1]) * s, (R[2][0] + R[0][2]) * s, (R[2][1] - R[1][2]) * s)
    if R[1][1] > R[0][0] and R[1][1] >= R[2][2]:
        s = math.sqrt(R[1][1] - R[2][2] - R[0][0] + 1.0); y = 0.5 * s; s = 0.5 / s
        return ((R[0][1] + R[1][0]) * s, y, (R[2][1] + R[1][2]) * s, (R[0][2] - R[2][0]) * s)
    s = math.sqrt(R[2][2] - R[0][0] - R[1][1] + 1.0); z = 0.5 * s; s = 0.5 / s
    return ((R[0][2] + R[2][0]) * s, (R[1][2] + R[2][1]) * s, z, (R[1][0] - R[0][1]) * s)


def se3_from_mat4f(m):
    """Converter::toSE3Quat(cv::Mat float 4x4)"""
    m = np.asarray(m, np.float32)
    R = [[float(m[r, c]) for c in range(3)] for r in range(3)]
    return se3_normalize((float(m[0, 3]), float(m[1, 3]), float(m[2, 3])), quat_from_R(R))


def pose7(a):
    return np.array(list(a[0]) + list(a[1]), np.float64)


def from_pose7(p):
    return ((float(p[0]), float(p[1]), float(p[2])), (float(p[3]), float(p[4]), float(p[5]), float(p[6])))


def zyx_euler_to_quat(roll, pitch, yaw):
    """matrix_utils.cc:18-31"""
    sy, cy = math.sin(yaw * 0.5), math.cos(yaw * 0.5)
    sp, cp = math.sin(pitch * 0.5), math.cos(pitch * 0.5)
    sr, cr = math.sin(roll * 0.5), math.cos(roll * 0.5)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


def detection_from_label(track_id, x1, y1, x2, y2, h, w, l, X, Y, Z, ry):
    """One `Car` line of ObjectTracking.txt as Tracking::ReadKittiObjectInfo (src/Tracking.cc:485-640) and the DetectionObject
    constructor (src/DetectionObject.cc:22-73) turn it into a detection: mrectBBox (cv::Rect of truncated doubles), mScale =
    (length, height, width), mTruthPosInCameraFrame = fromMinimalVector(X, Y - height / 2, Z, 0, ry, 0, scale)."""
    bx, by, bw, bh = float(x1), float(y1), float(x2) - float(x1), float(y2) - float(y1)
    scale = (float(l), float(h), float(w))
    q = zyx_euler_to_quat(0.0, float(ry), 0.0)                      # (mdRotZ = 0, mdRotY, mdRotX = 0) -> (roll, pitch, yaw)
    pose = se3_normalize((float(X), float(Y) - scale[1] / 2, float(Z)), q)
    return {"id": int(track_id), "bbox": (int(bx), int(by), int(bw), int(bh)), "scale": scale, "pose7": pose7(pose)}


def right_mask(mask):
    """Frame::ReadKittiSegmentationImage(folder, frame, rightseg = true) (src/Frame.cc:1217-1290) on the 8-bit id mask: every
    labelled pixel writes its label 49 pixels to both sides while the image is scanned in raster order, so a pixel ends up with the
    label of the rightmost labelled pixel within (x, x + 49] or, when there is none, with its own label (writes to the right are
    overwritten when the scan arrives there; column 0 is never written from the right: `if (j - k > 0)`)."""
    mask = np.asarray(mask, np.uint8)
    h, w = mask.shape
    cols = np.arange(w)
    idx = np.where(mask != 0, cols[None, :], -1)
    run = np.maximum.accumulate(idx, axis=1)                         # rightmost labelled column <= c
    r = run[:, np.minimum(cols + 49, w - 1)]                         # ... <= min(c + 49, w - 1)
    ok = (r > cols[None, :]) & (cols[None, :] > 0)
    out = mask.copy()
    rows, cc = np.nonzero(ok)
    out[rows, cc] = mask[rows, r[rows, cc]]
    return out


def object_masks(mask_l, mask_r):
    """LeftObjMask / RightObjMask of Frame::ExtractObjORB (src/Frame.cc:2632-2643): 255 on object pixels (label not 0, not 255)"""
    f = lambda m: np.where((m != 0) & (m != 255), 255, 0).astype(np.uint8)
    return f(mask_l), f(mask_r)


class CvRng:
    """cv::RNG with its default state (operator()(unsigned n) = next() % n; multiply-with-carry, core.hpp)"""

    def __init__(self):
        self.state = 0xFFFFFFFF

    def __call__(self, n):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return (self.state & 0xFFFFFFFF) % n


def _norm3(x, y, z):
    return math.sqrt(x * x + y * y + z * z)


class _ObjFrame:
    """what a Frame keeps per detection (mvObjKeysUn, mvObjPointsDescriptors, mvuObjKeysRight, mvObjPointDepth, mvObjKeysGrid,
    mvpMapObjectPoints, mvbObjKeysOutlier, mvDetectionObjects, mvMapObjects)"""


class ObjectTracker:
    """State of the tracking thread the object chain needs: the last frame's per-detection feature sets and the MapObjects."""

    def __init__(self, backend, K, bf, width, height, th_depth, grid, scale_factors, inv_level_sigma2):
        self.be = backend
        self.fx, self.fy, self.cx, self.cy = [F32(v) for v in K]
        self.bf = F32(bf)
        self.mb = F32(self.bf / self.fx)
        self.invfx = F32(1) / self.fx
        self.invfy = F32(1) / self.fy
        self.th_depth = F32(th_depth)            # mThDepth = mbf * ThDepth / fx, already converted by the caller
        self.w, self.h = width, height
        self.grid = grid
        self.sf = np.asarray(scale_factors, np.float32)
        self.is2 = np.asarray(inv_level_sigma2, np.float32)
        self.log_sf = F32(np.log(self.sf[1]))
        self.nlevels = len(self.sf)
        self.objects = {}                        # AllObjects by mnTruthID
        self.last = None                         # per-detection sets of the last frame
        self.frame_id = -1
        self.stats = []

    # ---- Frame::Frame, object part (after the two ORBextractor calls of the same frame) ----
    def make_frame(self, left, right, mask, detections):
        mask = np.ascontiguousarray(mask, np.uint8)
        mr = right_mask(mask)
        om_l, om_r = object_masks(mask, mr)
        kps, desc, ur, depth = self.be.extract_objects(left, right, om_l, om_r, self.mb, self.bf)
        F = _ObjFrame()
        F.dets = list(detections)
        n = len(F.dets)
        x = np.asarray(kps["x"], np.float32); y = np.asarray(kps["y"], np.float32)
        lab = mask[y.astype(np.int64), x.astype(np.int64)].astype(np.int64) if len(x) else np.zeros(0, np.int64)
        owner = np.full(len(x), -1, np.int64)
        for i in range(len(x)):
            if lab[i] != 0 and lab[i] != 255:
                for j, d in enumerate(F.dets):
                    oid = d["id"] - 255 if d["id"] > 255 else d["id"]
                    if oid == lab[i] - 1:
                        owner[i] = j
                        break
        F.obj = []
        for j in range(n):
            idx = np.nonzero(owner == j)[0]
            o = _ObjFrame()
            o.x, o.y = x[idx], y[idx]
            o.octave = np.asarray(kps["octave"], np.int32)[idx]
            o.angle = np.asarray(kps["angle"], np.float32)[idx]
            o.desc = np.asarray(desc, np.uint8).reshape(-1, 32)[idx]
            o.uright = np.asarray(ur, np.float32)[idx]
            o.depth = np.asarray(depth, np.float32)[idx]
            o.n = len(idx)
            o.cell_off, o.cell_idx = build_grid(o.x, o.y, *self.grid)
            o.mp_valid = np.zeros(o.n, bool)          # mvpMapObjectPoints[j][i] != NULL
            o.mp_observed = np.zeros(o.n, bool)       # ... ->Observations() > 0
            o.mp_id = np.full(o.n, -1, np.int64)      # index into the MapObject's point list (-1: temporal)
            o.mp_po = np.zeros((o.n, 3), np.float32)  # GetInObjFramePosition()
            o.outlier = np.zeros(o.n, np.uint8)
            o.mo = None                               # mvMapObjects[j]
            o.track_ok = False
            o.inliers = 0
            o.dynamic = True                          # DetectionObject::mbDynamicFlag (true from the constructor, DetectionObject.cc:74)
            o.dyn_mono = 0.0; o.dyn_stereo = 0.0      # mdMonoDynaVal / mdStereoDynaVal
            o.dyn_n = (0, 0)                          # points the two averages were taken over
            F.obj.append(o)
        F.n_temp = len(x)
        return F

    def _unproject(self, o, i):
        """Frame::UnprojectStereodynamic(order, i, false) (Frame.cc:2521-2544): camera-frame point, float arithmetic"""
        z = o.depth[i]
        xx = (o.x[i] - self.cx) * z * self.invfx
        yy = (o.y[i] - self.cy) * z * self.invfy
        return (F32(xx), F32(yy), F32(z))

    # ---- RANSAC centroid shared by InitializeCurrentObjPose / MapObjectInit / MapObjectReInit ----
    def _centroid(self, pts, fmax, iterations):
        l = len(pts)
        rng = CvRng()
        best, best_score = [], -1.0
        P = np.asarray(pts, np.float64).reshape(-1, 3)
        for _ in range(iterations):
            i1 = rng(l)
            dx = P[:, 0] - P[i1, 0]; dy = P[:, 1] - P[i1, 1]; dz = P[:, 2] - P[i1, 2]
            nrm = np.sqrt(dx * dx + dy * dy + dz * dz)
            inl = np.nonzero(nrm < float(fmax))[0]
            if float(len(inl)) > best_score:
                best_score = float(len(inl))
                best = inl
        return best

    def _mean(self, pts, inl):
        sx = sy = sz = 0.0
        for q in inl:
            sx += pts[q][0]; sy += pts[q][1]; sz += pts[q][2]
        m = float(len(inl))
        return [sx / m, sy / m, sz / m]

    def _project_rect(self, pose, scale):
        """ObjectState::projectOntoImageRectFromCamera (g2o_Object.cc:156-169) with EnObjectCenter = 0: [umin vmin umax vmax]"""
        R = quat_to_R(pose[1])
        t = pose[0]
        S = [[R[r][c] * (scale[c] * 0.5) for c in range(3)] for r in range(3)]
        body = ((1, 1, -1, -1, 1, 1, -1, -1), (1, -1, -1, 1, 1, -1, -1, 1), (-1, -1, -1, -1, 1, 1, 1, 1))
        fx, fy, cx, cy = float(self.fx), float(self.fy), float(self.cx), float(self.cy)
        lo = [0.0, 0.0]; hi = [0.0, 0.0]
        for k in range(8):
            c = [S[r][0] * body[0][k] + S[r][1] * body[1][k] + S[r][2] * body[2][k] + t[r] * 1.0 for r in range(3)]
            hw = 0.0 * body[0][k] + 0.0 * body[1][k] + 0.0 * body[2][k] + 1.0 * 1.0
            c = [v / hw for v in c]
            p0 = fx * c[0] + 0.0 * c[1] + cx * c[2]
            p1 = 0.0 * c[0] + fy * c[1] + cy * c[2]
            p2 = 0.0 * c[0] + 0.0 * c[1] + 1.0 * c[2]
            uv = (p0 / p2, p1 / p2)
            for r in range(2):
                lo[r] = uv[r] if k == 0 else min(lo[r], uv[r])
                hi[r] = uv[r] if k == 0 else max(hi[r], uv[r])
        return (lo[0], lo[1], hi[0], hi[1])

    def _fine_tune(self, det, pose, scale):
        """Tracking::FineTuningUsing2dBox (src/Tracking.cc:1704-1786): the cuboid's projection is aligned with the detection box
        (centre row, height when farther than 8 m, centre column) by stepping the translation.  cv::Rect / cv::Point are int."""
        bx, by, bw, bh = det["bbox"]
        rc = ((bx + (bx + bw)) // 1, (by + (by + bh)) // 1)        # tl + br
        idiv = lambda a: int(a / 2)                                  # Point_<int> / 2: integer division, truncating
        rcx, rcy = idiv(rc[0]), idiv(rc[1])

        def box(p):
            pr = self._project_rect(p, scale)
            x0, y0 = int(pr[0]), int(pr[1])
            return x0, y0, int(pr[2] - pr[0]), int(pr[3] - pr[1])

        t = list(pose[0]); q = pose[1]
        pb = box((t, q))
        dcx = idiv(pb[0] + pb[0] + pb[2]) - rcx; dcy = idiv(pb[1] + pb[1] + pb[3]) - rcy
        for _ in range(400):
            direction = -1 if dcy < 0 else 1
            t[1] = t[1] - direction * 0.01
            pb = box((t, q))
            dcx = idiv(pb[0] + pb[0] + pb[2]) - rcx; dcy = idiv(pb[1] + pb[1] + pb[3]) - rcy
            if abs(dcy) < 1:
                break
        if t[2] > 8:
            dh = pb[3] - bh
            for _ in range(400):
                direction = -1 if dh < 0 else 1
                t[2] = t[2] + direction * 0.05
                pb = box((t, q))
                dh = pb[3] - bh
                if abs(dh) < 1:
                    break
        dcx = idiv(pb[0] + pb[0] + pb[2]) - rcx
        for _ in range(400):
            direction = -1 if dcx < 0 else 1
            t[0] = t[0] - direction * 0.01
            pb = box((t, q))
            dcx = idiv(pb[0] + pb[0] + pb[2]) - rcx
            if abs(dcx) < 1:
                break
        return (tuple(t), q)

    def _fmax(self, scale):
        return F32(_norm3(scale[0], scale[1], scale[2]))           # float fMaxDis = scale.norm()

    def _cam_points(self, o, order=None):
        idx = [i for i in range(o.n) if o.depth[i] > 0] if order is None else order
        return [tuple(float(v) for v in self._unproject(o, i)) for i in idx], idx

    def _keyframe_points(self, o, mo, pose, inl, pts, idx, fmax, init):
        """the MapObjectPoints of a new ObjectKeyFrame (MapObjectInit / MapObjectReInit tail): object-frame position, one
        observation, descriptor of the keypoint, UpdateNormalAndDepth against the keyframe's camera centre"""
        inv = se3_inverse(pose)
        m = np.zeros((4, 4), np.float32)
        R = quat_to_R(pose[1])
        for r in range(3):
            for c in range(3):
                m[r, c] = F32(R[r][c])
            m[r, 3] = F32(pose[0][r])
        # ObjectKeyFrame::SetPose: mPoc = -Roc * tco (float)
        poc = [F32(-(float(m[0, r]) * float(m[0, 3]) + float(m[1, r]) * float(m[1, 3]) + float(m[2, r]) * float(m[2, 3]))) for r in range(3)]
        rows = []
        for j in inl:
            x3do = se3_map(inv, pts[j])
            if _norm3(*x3do) > float(fmax):
                continue
            n = idx[j]
            po = (F32(x3do[0]), F32(x3do[1]), F32(x3do[2]))
            v = (po[0] - poc[0], po[1] - poc[1], po[2] - poc[2])
            nd = math.sqrt(float(v[0]) * float(v[0]) + float(v[1]) * float(v[1]) + float(v[2]) * float(v[2]))
            dist = F32(nd)
            inv_n = F32(1.0 / nd)
            maxd = F32(dist * self.sf[o.octave[n]])
            rows.append((n, po, (v[0] * inv_n, v[1] * inv_n, v[2] * inv_n), maxd, F32(maxd / self.sf[self.nlevels - 1])))
        # the keyframe's GetMapObjectPointMatches() - the object's local map - lists the points by feature index
        rows.sort(key=lambda r: r[0])
        k = len(rows)
        P = {"po": np.zeros((k, 3), np.float32), "normal": np.zeros((k, 3), np.float32), "max_dist": np.zeros(k, np.float32),
             "min_dist": np.zeros(k, np.float32), "desc": np.zeros((k, 32), np.uint8)}
        for pid, (n, po, nrm, maxd, mind) in enumerate(rows):
            P["po"][pid] = po; P["normal"][pid] = nrm; P["max_dist"][pid] = maxd; P["min_dist"][pid] = mind; P["desc"][pid] = o.desc[n]
            o.mp_valid[n] = True; o.mp_observed[n] = True; o.mp_id[n] = pid; o.mp_po[n] = po; o.outlier[n] = 0
        mo["points"] = P
        if init:                                     # mnLastKeyFrameId: MapObjectInit (Tracking.cc:1875), never MapObjectReInit (:1908-2031)
            mo["kf_frame"] = self.frame_id
        mo["local_valid"] = False                    # mvLocalObjectKeyFrames is filled by the first UpdateObjectLocalKeyFrames that finds observations

    # ---- Tracking::MapObjectInit ----
    def _map_object_init(self, F, j):
        o, det = F.obj[j], F.dets[j]
        scale = det["scale"]
        fmax = self._fmax(scale)
        pts, idx = self._cam_points(o)
        l = len(pts)
        inl = self._centroid(pts, fmax, int(0.8 * l)) if l else []
        if len(inl) < 3:
            return
        c = self._mean(pts, inl)
        if c[2] < 8:
            return
        c[2] += 0.2 * scale[0]
        c[1] = 0 + scale[1] / 2
        truth = from_pose7(det["pose7"])
        pose = self._fine_tune(det, (tuple(c), truth[1]), scale)
        # new MapObject(id, candidate_cuboid->GetDynamicFlag(), ...): mbDynamicChanged(false), mbFirstObserved(true), empty history (MapObject.cc:20-21)
        mo = {"id": det["id"], "first_frame": self.frame_id, "scale": scale, "tco": pose, "tco_frame": self.frame_id,
              "dynamic": bool(o.dynamic), "dyn_changed": False, "dyn_first": True, "dyn_hist": []}
        self.objects[det["id"]] = mo
        o.mo = mo
        self._keyframe_points(o, mo, pose, inl, pts, idx, fmax, True)
        o.new = True

    # ---- Tracking::MapObjectReInit ----
    def _map_object_reinit(self, F, j):
        o, det = F.obj[j], F.dets[j]
        mo = o.mo
        scale = det["scale"]
        # pMO->ClearMapObjectPoint(): the abandoned keyframe and its points leave the slice (the reference leaves the stale points
        # reachable through the frame's pointers until ObjectLocalMapping culls them)
        mo["points"] = {k: v[:0] for k, v in mo["points"].items()}
        o.mp_valid[:] = False; o.mp_observed[:] = False; o.mp_id[:] = -1; o.outlier[:] = 0
        fmax = self._fmax(scale)
        cand = [i for i in range(o.n) if o.depth[i] > 0]
        cand.sort(key=lambda i: (float(o.depth[i]), i))
        order = []
        for i in cand:
            order.append(i)
            if o.depth[i] > 2 * self.th_depth and len(order) > 100:
                break
        pts, idx = self._cam_points(o, order)
        l = len(pts)
        inl = self._centroid(pts, fmax, l) if l else []
        if len(inl) <= 3:
            return
        c = self._mean(pts, inl)
        if c[2] < 8:
            return
        if c[2] > 8:
            c[2] += 0.2 * scale[0]
        c[1] = 0 + scale[1] / 2
        truth = from_pose7(det["pose7"])
        pose = self._fine_tune(det, (tuple(c), truth[1]), scale)
        mo["tco"] = pose; mo["tco_frame"] = self.frame_id
        self._keyframe_points(o, mo, pose, inl, pts, idx, fmax, False)

    # ---- Tracking::TrackMapObject ----
    def _track_map_object(self, F, tcl):
        F.in_last = []       # mvInLastFrameTrackedObjOrders: (order in the last frame, order in this frame)
        F.tracked = []       # mvTotalTrackedObjOrders
        for j, det in enumerate(F.dets):
            o = F.obj[j]
            o.new = False
            mo = self.objects.get(det["id"])
            if mo is None:
                self._map_object_init(F, j)
                continue
            scale = mo["scale"]
            pose = se3_mul(tcl, mo["tco"])
            # InitializeCurrentObjPose: RANSAC centroid of the detection's stereo points replaces the translation
            pts, idx = self._cam_points(o)
            l = len(pts)
            fmax = self._fmax(det["scale"])
            inl = self._centroid(pts, fmax, int(0.8 * l)) if l else []
            if len(inl) >= 3:
                c = self._mean(pts, inl)
                if c[2] > 8:
                    c[2] += 0.2 * det["scale"][0]
                c[1] = 0 + det["scale"][1] / 2
                pose = (tuple(c), pose[1])
            pose = self._fine_tune(det, pose, scale)
            latest = mo["tco_frame"]
            mo["tco"] = pose; mo["tco_frame"] = self.frame_id
            o.mo = mo
            if latest == self.frame_id - 1 and self.last is not None:
                lj = next((k for k, d in enumerate(self.last.dets) if d["id"] == det["id"]), -1)
                assert lj >= 0
                F.in_last.append((lj, j))
            else:
                o.dynamic = mo["dynamic"]            # candidate_cuboid->SetDynamicFlag(object_temp->GetDynamicFlag()) (Tracking.cc:1617)
            F.tracked.append(j)

    def _cfse3(self, F, orders):
        objs = []
        for j in orders:
            o = F.obj[j]
            objs.append({"xo": o.mp_po, "obs": np.stack([o.x, o.y, o.uright], 1).astype(np.float32) if o.n else np.zeros((0, 3), np.float32),
                         "inv_sigma2": self.is2[o.octave], "valid": o.mp_valid.astype(np.uint8), "pose7": pose7(o.mo["tco"])})
        ok, poses, outl = self.be.cfse3(objs, (self.fx, self.fy, self.cx, self.cy, self.bf))
        if ok:
            for k, j in enumerate(orders):
                o = F.obj[j]
                o.mo["tco"] = from_pose7(poses[k])
                o.outlier = np.asarray(outl[k], np.uint8).copy()
        return ok

    # ---- Tracking::TrackLastFrameObjectPoint ----
    def _track_last_frame(self, F):
        L = self.last
        for lj, j in F.in_last:
            lo = L.obj[lj]
            mo = lo.mo
            if mo["kf_frame"] == self.frame_id - 1 or mo["first_frame"] == self.frame_id - 1:
                continue
            cand = [i for i in range(lo.n) if lo.depth[i] > 0]
            if not cand:
                continue
            cand.sort(key=lambda i: (float(lo.depth[i]), i))
            tco_last = lo.tco_at_frame
            inv = se3_inverse(tco_last)
            fmax = self._fmax(mo["scale"])
            for i in cand:
                if not lo.mp_valid[i] or not lo.mp_observed[i]:
                    pc = self._unproject(lo, i)
                    po = se3_map(inv, tuple(float(v) for v in pc))
                    pf = (F32(po[0]), F32(po[1]), F32(po[2]))
                    nf = np.sqrt(pf[0] * pf[0] + pf[1] * pf[1] + pf[2] * pf[2])       # Eigen::Vector3f::norm()
                    if nf > fmax:
                        continue                                                 # (the reference `continue`s past the break test as well)
                    lo.mp_valid[i] = True; lo.mp_observed[i] = False; lo.mp_id[i] = -1; lo.mp_po[i] = pf
                if lo.depth[i] > 2 * self.th_depth:
                    break
        need = []
        probs = []
        for lj, j in F.in_last:
            lo, o = L.obj[lj], F.obj[j]
            probs.append({"q_desc": lo.desc, "q_angle": lo.angle, "q_valid": (lo.mp_valid & (lo.outlier == 0)).astype(np.uint8),
                          "t_desc": o.desc, "t_angle": o.angle})
        res = self.be.search_bruteforce(probs) if probs else []
        for n, (lj, j) in enumerate(F.in_last):
            lo, o = L.obj[lj], F.obj[j]
            nm, qot = res[n]
            m = qot >= 0
            o.mp_valid[:] = m
            o.mp_observed[:] = False; o.mp_id[:] = -1
            o.mp_observed[m] = lo.mp_observed[qot[m]]
            o.mp_id[m] = lo.mp_id[qot[m]]
            o.mp_po[m] = lo.mp_po[qot[m]]
            o.bf_matches = int(nm)
            if nm >= 10:
                need.append(j)
        if not need:
            return
        self._cfse3(F, need)
        for j in need:
            o = F.obj[j]
            drop = o.mp_valid & (o.outlier != 0)
            o.mp_valid[drop] = False
            o.outlier[drop] = 0
            o.track_ok = int((o.mp_valid & o.mp_observed).sum()) >= 10

    # ---- Tracking::TrackObjectLocalMap (UpdateObjectLocalKeyFrames / Points, SearchObjectLocalPoints) ----
    def _track_local_map(self, F):
        need = []
        probs, pidx = [], []
        for j in F.tracked:
            o = F.obj[j]
            mo = o.mo
            P = mo["points"]
            npts = len(P["po"])
            # UpdateObjectLocalKeyFrames: the keyframes observing the frame's points; without any the list keeps its last content
            if bool((o.mp_valid & o.mp_observed).any()):
                mo["local_valid"] = True
            nloc = npts if mo["local_valid"] else 0
            seen = np.zeros(max(npts, 1), bool)
            ids = o.mp_id[o.mp_id >= 0]                        # mnLastFrameSeen == mCurrentFrame.mnId: the frame's points and the
            seen[ids] = True                                   # outliers the first CFSE3 just discarded (Tracking.cc:2441-2444)
            bx, by, bw, bh = F.dets[j]["bbox"]
            tco = mo["tco"]
            inv = se3_inverse(tco)
            poc = (F32(inv[0][0]), F32(inv[0][1]), F32(inv[0][2]))
            q = {"valid": np.zeros(nloc, np.uint8), "proj_x": np.zeros(nloc, np.float32), "proj_y": np.zeros(nloc, np.float32),
                 "proj_xr": np.zeros(nloc, np.float32), "level": np.zeros(nloc, np.int32), "view_cos": np.zeros(nloc, np.float32),
                 "desc": P["desc"][:nloc], "observed": np.ones(nloc, np.uint8)}
            nto = 0
            for i in range(nloc):
                if seen[i]:
                    continue
                po = P["po"][i]
                pc = se3_map(tco, (float(po[0]), float(po[1]), float(po[2])))
                X, Y, Z = F32(pc[0]), F32(pc[1]), F32(pc[2])
                if Z < 0:
                    continue
                invz = F32(1) / Z
                u = self.fx * X * invz + self.cx
                v = self.fy * Y * invz + self.cy
                if not (float(u) >= float(bx) and float(u) < float(bx) + float(bw) and float(v) >= float(by) and float(v) < float(by) + float(bh)):
                    continue
                d = (po[0] - poc[0], po[1] - poc[1], po[2] - poc[2])
                dist = F32(math.sqrt(float(d[0]) * float(d[0]) + float(d[1]) * float(d[1]) + float(d[2]) * float(d[2])))
                maxd = F32(1.2) * P["max_dist"][i]; mind = F32(0.8) * P["min_dist"][i]
                if dist < mind or dist > maxd:
                    continue
                pn = P["normal"][i]
                dot = float(d[0]) * float(pn[0]) + float(d[1]) * float(pn[1]) + float(d[2]) * float(pn[2])
                view_cos = F32(dot / float(dist))
                if view_cos < F32(0.5):
                    continue
                ratio = P["max_dist"][i] / dist
                level = int(math.ceil(F32(F32(math.log(float(ratio)))) / self.log_sf))
                level = 0 if level < 0 else (self.nlevels - 1 if level >= self.nlevels else level)
                q["valid"][i] = 1; q["proj_x"][i] = u; q["proj_y"][i] = v; q["proj_xr"][i] = u - self.bf * invz
                q["level"][i] = level; q["view_cos"][i] = view_cos
                nto += 1
            o.lm_candidates = nto
            if nto > 0:
                in_bbox = ((o.x.astype(np.float64) >= bx) & (o.x.astype(np.float64) < float(bx) + float(bw)) &
                           (o.y.astype(np.float64) >= by) & (o.y.astype(np.float64) < float(by) + float(bh))).astype(np.uint8)
                train = {"x": o.x, "y": o.y, "octave": o.octave, "angle": o.angle, "u_right": o.uright, "desc": o.desc,
                         "occupied": (o.mp_valid & o.mp_observed).astype(np.uint8), "in_bbox": in_bbox, "grid": self.grid,
                         "cell_off": o.cell_off, "cell_idx": o.cell_idx}
                probs.append({"mode": "points", "object": True, "train": train, "query": q, "scale_factors": self.sf, "th": 1.0})
                pidx.append(j)
            need.append(j)
        res = self.be.search_object_points(probs) if probs else []
        for (nm, match), j in zip(res, pidx):
            o = F.obj[j]
            P = o.mo["points"]
            m = match >= 0
            o.mp_valid[m] = True; o.mp_observed[m] = True; o.mp_id[m] = match[m]
            o.mp_po[m] = P["po"][match[m]]
            o.lm_matches = int(nm)
        if not need:
            return
        self._cfse3(F, need)
        for j in need:
            o = F.obj[j]
            good = o.mp_valid & (o.outlier == 0)
            o.inliers = int((good & o.mp_observed).sum())
            o.mp_valid[o.mp_valid & (o.outlier != 0)] = False
            o.track_ok = o.inliers > 10

    # ---- MapObject::DynamicDetection + SetDynamicFlag (MapObject.cc:414-448) ----
    @staticmethod
    def _mo_dynamic_detection(mo, flag):
        if not mo["dyn_changed"]:
            h = mo["dyn_hist"]
            h.append(bool(flag))
            if len(h) >= 4:
                if len(h) > 4:
                    h.pop(0)
                if all(v == bool(flag) for v in h) and mo["dynamic"] != bool(flag):
                    mo["dyn_changed"] = True

    @staticmethod
    def _mo_set_dynamic(mo, flag):
        if mo["dyn_first"]:
            mo["dynamic"] = bool(flag); mo["dyn_first"] = False
        if mo["dyn_changed"]:
            mo["dynamic"] = bool(flag); mo["dyn_changed"] = False

    # ---- Tracking::DynamicStaticDiscrimination (Tracking.cc:2058-2202) without its tail StaticPointRecoveryFromObj ----
    def _dynamic_static_discrimination(self, F, tcw_cur, tcw_last):
        ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))
        # mCurrentFrame.mSETcw / mLastFrame.mSETcw = Converter::toSE3Quat(mTcw) (Frame.cc:1669); a frame without a pose keeps the
        # default-constructed SE3Quat (identity)
        cur = se3_from_mat4f(tcw_cur) if tcw_cur is not None else ident
        last = se3_from_mat4f(tcw_last) if tcw_last is not None else ident
        jobs, who = [], []
        for lj, j in F.in_last:
            o, det, mo = F.obj[j], F.dets[j], F.obj[j].mo
            depth = mo["tco"][0][2]
            if depth < 7 or depth > float(self.th_depth):
                o.dynamic = mo["dynamic"]
                continue
            bx, by, bw, bh = det["bbox"]
            middle_x = float(self.w // 2)                       # double middle_x = mImGray.size[1] / 2  (int / int)
            current_px = float(bx + int(bw / 2))                # cv::Rect members are int
            if abs(current_px - middle_x) < int(bw / 2) + 60:   # image prior: a vehicle straight ahead moves
                o.dynamic = True
                self._mo_set_dynamic(mo, True)
                continue
            jobs.append({"valid": o.mp_valid.astype(np.uint8), "po": o.mp_po.astype(np.float64),
                         "obs": np.stack([o.x, o.y, o.uright], 1).astype(np.float32) if o.n else np.zeros((0, 3), np.float32),
                         "inv_sigma2": self.is2[o.octave], "last_tco": pose7(self.last.obj[lj].tco_at_frame), "last_tcw": pose7(last),
                         "cur_tcw": pose7(cur), "K": (float(self.fx), float(self.fy), float(self.cx), float(self.cy)), "mbf": self.bf})
            who.append(j)
        res = self.be.dynamic_discrimination(jobs) if jobs else []
        for (mono, stereo, n_mono, n_stereo), j in zip(res, who):
            o, mo = F.obj[j], F.obj[j].mo
            o.dyn_n = (int(n_mono), int(n_stereo))
            if mono > 0 or stereo > 0:                          # DetectionObject::SetDynamicFlag(mono, stereo) (DetectionObject.cc:169-196)
                o.dyn_mono, o.dyn_stereo = float(mono), float(stereo)
                o.dynamic = bool(mono > 1 or stereo > 2)
                self._mo_dynamic_detection(mo, o.dynamic)
                self._mo_set_dynamic(mo, o.dynamic)
            else:
                o.dynamic = mo["dynamic"]

    # ---- the object part of Tracking::Track for one frame (camera poses of the last and the current frame: 4x4 float or None) ----
    def track(self, left, right, mask, detections, tcw_cur, tcw_last, camera_initialized):
        self.frame_id += 1
        F = self.make_frame(left, right, mask, detections)
        st = {"n_temp": F.n_temp, "objects": []}
        if not camera_initialized:
            # the frame of StereoInitialization: Track returns before the object functions; the frame still becomes mLastFrame
            self._finish(F, st)
            return st
        if tcw_cur is not None and tcw_last is not None:
            from .tracker import mul4, inverse_rt
            tcl = se3_from_mat4f(mul4(tcw_cur, inverse_rt(tcw_last)))        # camera_Tcl = mCurrentFrame.mTcw * mLastFrame.mTwc
        else:
            tcl = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))
        self._track_map_object(F, tcl)
        if F.in_last:
            self._track_last_frame(F)
        if F.tracked:
            self._track_local_map(F)
        self._dynamic_static_discrimination(F, tcw_cur, tcw_last)      # Moving Objects Recognition (Tracking.cc:1244)
        # end of Track, SLOT mode 4 (Tracking.cc:1443-1478)
        for j, o in enumerate(F.obj):
            if o.mo is None or o.mo["first_frame"] == self.frame_id:
                continue
            tmp = o.mp_valid & ~o.mp_observed
            o.mp_valid[tmp] = False
            o.outlier[tmp] = 0
            if not o.track_ok:
                self._map_object_reinit(F, j)
        self._finish(F, st)
        return st

    def _finish(self, F, st):
        for j, o in enumerate(F.obj):
            o.tco_at_frame = o.mo["tco"] if o.mo is not None else None
            st["objects"].append({"id": F.dets[j]["id"], "n": o.n, "stereo": int((o.depth > 0).sum()), "tracked": o.mo is not None,
                                  "new": bool(getattr(o, "new", False)), "track_ok": bool(o.track_ok), "inliers": int(o.inliers),
                                  "bf_matches": int(getattr(o, "bf_matches", 0)), "lm_candidates": int(getattr(o, "lm_candidates", 0)),
                                  "lm_matches": int(getattr(o, "lm_matches", 0)), "map_points": int((o.mp_valid & o.mp_observed).sum()),
                                  "tco": None if o.mo is None else pose7(o.mo["tco"]),
                                  "dynamic": bool(o.dynamic), "mo_dynamic": None if o.mo is None else bool(o.mo["dynamic"]),
                                  "dyn_mono": float(o.dyn_mono), "dyn_stereo": float(o.dyn_stereo), "dyn_n": tuple(o.dyn_n)})
        self.last = F
        self.stats.append(st)
