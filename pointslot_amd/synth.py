"""Seeded synthetic inputs for the BASELINE.json configs.  Pure numpy and
self-contained, so every consumer (GPU path, CPU checker, bench) sees identical bytes.

RNG: xoshiro256** run as LANES independent lock-step streams (each lane seeded by splitmix64 from
(seed, lane)); values are consumed step-major.  Deterministic for a given (seed, lanes).
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return x, z ^ (z >> np.uint64(31))


def _rotl(x, k):
    return ((x << np.uint64(k)) | (x >> np.uint64(64 - k))) & _M


class Rng:
    def __init__(self, seed, lanes=256):
        with np.errstate(over="ignore"):
            x = (np.uint64(seed) + np.arange(lanes, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)) & _M
            s = []
            for _ in range(4):
                x, z = _splitmix64(x)
                s.append(z)
        self.s = s
        self.lanes = lanes
        self._buf = np.zeros(0, np.uint64)

    def _step(self):
        with np.errstate(over="ignore"):
            s0, s1, s2, s3 = self.s
            r = (_rotl((s1 * np.uint64(5)) & _M, 7) * np.uint64(9)) & _M
            t = (s1 << np.uint64(17)) & _M
            s2 = s2 ^ s0
            s3 = s3 ^ s1
            s1 = s1 ^ s2
            s0 = s0 ^ s3
            s2 = s2 ^ t
            s3 = _rotl(s3, 45)
            self.s = [s0, s1, s2, s3]
        return r

    def u64(self, n):
        while self._buf.size < n:
            self._buf = np.concatenate([self._buf, self._step()])
        out, self._buf = self._buf[:n], self._buf[n:]
        return out

    def uniform(self, n, lo=0.0, hi=1.0):
        u = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def integers(self, n, lo, hi):
        """uniform integers in [lo, hi)"""
        return (lo + np.floor(self.uniform(n) * (hi - lo))).astype(np.int64)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return z[:n]


# ---- config 2: stereo pair for the ORB extractor ---------------------------------------------------
def _value_noise(rng, w, h, cell, amp):
    gw, gh = w // cell + 2, h // cell + 2
    g = rng.uniform(gw * gh, -1.0, 1.0).reshape(gh, gw)
    ys, xs = np.arange(h) / cell, np.arange(w) / cell
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def stereo_pair(seed=0x51070002, w=1242, h=375, n_rect=400, bf=384.38148):
    """Value-noise texture (amplitude 64/32/16 on lattices of 32/16/8 px, plus a fine 3-px octave of
    amplitude 14 so that FAST sees a realistic few thousand candidates per level) plus `n_rect`
    uniform grey rectangles of 8..64 px for corners; right image = left warped by the disparity
    d(y) = bf / z(y), z linear 6 m (bottom) .. 60 m (top), bilinear.  Returns two uint8 (h, w) arrays."""
    rng = Rng(seed)
    img = np.full((h, w), 128.0)
    for cell, amp in ((32, 64.0), (16, 32.0), (8, 16.0), (3, 14.0)):
        img += _value_noise(rng, w, h, cell, amp)
    rw = rng.integers(n_rect, 8, 65); rh = rng.integers(n_rect, 8, 65)
    rx = rng.integers(n_rect, 0, w); ry = rng.integers(n_rect, 0, h)
    rg = rng.integers(n_rect, 0, 256)
    for i in range(n_rect):
        img[ry[i]:ry[i] + rh[i], rx[i]:rx[i] + rw[i]] = rg[i]
    left = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    z = 60.0 + (6.0 - 60.0) * (np.arange(h) / (h - 1))
    d = bf / z
    xs = np.arange(w)[None, :] + d[:, None]          # right(x) = left(x + d)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    fx = xs - np.floor(xs)
    rows = np.arange(h)[:, None]
    lf = left.astype(np.float64)
    right = np.clip(np.rint(lf[rows, x0] * (1 - fx) + lf[rows, x1] * fx), 0, 255).astype(np.uint8)
    return left, right


def stereo_batch(n_pairs, seed=0x51070002, w=1242, h=375):
    """`n_pairs` stereo pairs (seed + k) as one uint8 array [2 * n_pairs, h, w] (L0, R0, L1, R1, ...)."""
    out = np.zeros((2 * n_pairs, h, w), np.uint8)
    for k in range(n_pairs):
        l, r = stereo_pair(seed + k, w, h)
        out[2 * k], out[2 * k + 1] = l, r
    return out


# ---- brute-force matching problems (object feature sets of two consecutive frames) ----------------------
def bruteforce_problem(seed, nq=300, nt=320, p_same=0.7, flip_bits=18, p_valid=0.9, dup_frac=0.1):
    """nq query descriptors; a fraction p_same of them re-appear among the nt train descriptors with
    ~flip_bits random bit flips (so the best distance is below TH_LOW) and a consistent rotation; a
    fraction dup_frac of the trains are near-duplicates of other trains so that the ratio test and the
    'train already taken' rule are exercised."""
    rng = Rng(seed)
    qd = rng.integers(nq * 32, 0, 256).astype(np.uint8).reshape(nq, 32)
    td = rng.integers(nt * 32, 0, 256).astype(np.uint8).reshape(nt, 32)
    qa = rng.uniform(nq, 0.0, 360.0).astype(np.float32)
    ta = rng.uniform(nt, 0.0, 360.0).astype(np.float32)
    perm = np.argsort(rng.uniform(nt))
    nsame = min(int(nq * p_same), nt)
    src = np.argsort(rng.uniform(nq))[:nsame]
    for k in range(nsame):
        j, i = perm[k], src[k]
        bits = np.unpackbits(qd[i])
        nflip = int(rng.integers(1, 0, flip_bits + 1)[0])
        pos = rng.integers(nflip, 0, 256)
        bits[pos] ^= 1
        td[j] = np.packbits(bits)
        rot = 25.0 + rng.normal(1)[0] * (3.0 if rng.uniform(1)[0] < 0.85 else 60.0)
        ta[j] = np.float32((qa[i] - rot) % 360.0)
    ndup = int(nt * dup_frac)
    a = rng.integers(ndup, 0, nt); b = rng.integers(ndup, 0, nt)
    for k in range(ndup):
        bits = np.unpackbits(td[a[k]])
        bits[rng.integers(3, 0, 256)] ^= 1
        td[b[k]] = np.packbits(bits)
    qv = (rng.uniform(nq) < p_valid).astype(np.uint8)
    return {"q_desc": qd, "q_angle": qa, "q_valid": qv, "t_desc": td, "t_angle": ta}


# ---- config 3: per-frame pose optimisation problems ----------------------------------------------------
KITTI_K = (721.5377, 721.5377, 609.5593, 172.8540)   # fx, fy, cx, cy  (Examples/Stereo/0000-0013.yaml:8-11)
KITTI_BF = 384.38148                                  # Camera.bf (yaml:26)
_QUOTAS = np.array([434, 362, 302, 251, 209, 175, 145, 122], np.float64)


def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def _level_sigma():
    s = [np.float32(1.0)]
    for _ in range(7):
        s.append(np.float32(np.float64(s[-1]) * np.float64(np.float32(1.2))))
    s = np.array(s, np.float32)
    return s, (np.float32(1.0) / (s * s)).astype(np.float32)


def pose_problem(seed, n=2000, outlier_frac=0.1, noise=1.0, mono_frac=0.0, valid_frac=1.0, w=1242, h=375):
    """One PoseOptimization input: n map points in the frustum (z in [5,60] m), true
    Tcw = exp(xi) with rot within +-3 deg and trans within +-0.5 m, initial guess identity, octave ~ level
    quotas, pixel noise sigma = noise * 1.2^octave on (u, v, uR), a fraction of uniform-in-image outliers."""
    rng = Rng(seed)
    fx, fy, cx, cy = KITTI_K
    R = _so3_exp(np.radians(rng.uniform(3, -3.0, 3.0)))
    t = rng.uniform(3, -0.5, 0.5)
    u = rng.uniform(n, 0, w); v = rng.uniform(n, 0, h); z = rng.uniform(n, 5.0, 60.0)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = ((Xc - t) @ R).astype(np.float32)                      # R^T (Xc - t)
    Xc = Xw.astype(np.float64) @ R.T + t                        # re-project the float32 points
    u = fx * Xc[:, 0] / Xc[:, 2] + cx; v = fy * Xc[:, 1] / Xc[:, 2] + cy; ur = u - KITTI_BF / Xc[:, 2]
    octave = np.searchsorted(np.cumsum(_QUOTAS) / _QUOTAS.sum(), rng.uniform(n)).clip(0, 7)
    sig, inv_sigma2 = _level_sigma()
    sd = noise * sig[octave].astype(np.float64)
    obs = np.stack([u + sd * rng.normal(n), v + sd * rng.normal(n), ur + sd * rng.normal(n)], 1)
    out = rng.uniform(n) < outlier_frac
    no = int(out.sum())
    obs[out, 0] = rng.uniform(no, 0, w); obs[out, 1] = rng.uniform(no, 0, h)
    obs[out, 2] = obs[out, 0] - rng.uniform(no, 1.0, 60.0)
    mono = rng.uniform(n) < mono_frac
    obs[mono, 2] = -1.0
    valid = (rng.uniform(n) < valid_frac).astype(np.uint8)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return {"xw": Xw, "obs": obs.astype(np.float32), "inv_sigma2": inv_sigma2[octave], "valid": valid,
            "K": (np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), np.float32(KITTI_BF)),
            "tcw0": np.eye(4, dtype=np.float32), "tcw_true": T, "is_outlier": out, "octave": octave}


# ---- config 4: object local bundle adjustment ------------------------------------------------------------
def _quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = (R[2, 1] - R[1, 2]) / (4 * w); y = (R[0, 2] - R[2, 0]) / (4 * w); z = (R[1, 0] - R[0, 1]) / (4 * w)
    return np.array([x, y, z, w])


def _Ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _Rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def object_ba_problem(seed, n_kf=50, n_pts=300, p_vis=1.0, outlier_frac=0.05, noise=1.0, n_fixed_extra=0,
                      perturb=(0.3, 5.0, 0.1), perturb_axis="y", mono_frac=0.0):
    """One ObjectLocalBundleAdjustment graph: a 4.0 x 1.6 x 1.5 m cuboid with n_pts
    points, n_kf object keyframes whose Tco follows a path (z 8 -> 30 m, x +-4 m, yaw sweep +-30 deg), every
    (KF, point) visible with probability p_vis, pixel noise sigma = noise * 1.2^octave, outliers, KF 0 fixed,
    the others VertexSE3Fix{roll/pitch fixed}; n_fixed_extra extra fixed observers (VertexSE3Expmap fixed).
    perturb = (metres, degrees, metres) for pose translation / rotation / points; perturb_axis 'y' follows the
    survey (yaw about the camera's y), 'z' perturbs the only rotation the reference's vertex can correct."""
    rng = Rng(seed)
    fx, fy, cx, cy = KITTI_K
    pts = np.stack([rng.uniform(n_pts, -2.0, 2.0), rng.uniform(n_pts, -0.8, 0.8), rng.uniform(n_pts, -0.75, 0.75)], 1)
    P = n_kf + n_fixed_extra
    s = np.linspace(0, 1, P)
    yaw = np.radians(-30 + 60 * s)
    tz = 8 + 22 * s; tx = 4 * np.sin(2 * np.pi * s); ty = np.full(P, 1.0)
    poses_true = np.zeros((P, 7)); poses_init = np.zeros((P, 7))
    flags = np.zeros(P, np.uint8)
    sig, inv_sigma2 = _level_sigma()
    e_pose, e_point, e_obs, e_is2, e_out = [], [], [], [], []
    for i in range(P):
        R = _Ry(yaw[i]); t = np.array([tx[i], ty[i], tz[i]])
        poses_true[i, :3] = t; poses_true[i, 3:] = _quat_from_R(R)
        if i == 0 or i >= n_kf:
            flags[i] = 1 | (2 if i == 0 else 0)
            Ri, ti = R, t
        else:
            flags[i] = 2
            ang = np.radians(rng.uniform(1, -perturb[1], perturb[1])[0])
            Ri = (_Ry(ang) if perturb_axis == "y" else _Rz(ang)) @ R
            ti = t + rng.uniform(3, -perturb[0], perturb[0])
        poses_init[i, :3] = ti; poses_init[i, 3:] = _quat_from_R(Ri)
        Xc = pts @ R.T + t
        vis = rng.uniform(n_pts) < p_vis
        octave = np.clip(np.floor(np.log(np.maximum(Xc[:, 2], 1e-3) / 8.0) / np.log(1.2)).astype(int), 0, 7)
        sd = noise * sig[octave].astype(np.float64)
        u = fx * Xc[:, 0] / Xc[:, 2] + cx + sd * rng.normal(n_pts)
        v = fy * Xc[:, 1] / Xc[:, 2] + cy + sd * rng.normal(n_pts)
        ur = fx * Xc[:, 0] / Xc[:, 2] + cx - KITTI_BF / Xc[:, 2] + sd * rng.normal(n_pts)
        isout = rng.uniform(n_pts) < outlier_frac
        du = rng.uniform(n_pts, -40, 40); dv = rng.uniform(n_pts, -40, 40)
        u = np.where(isout, u + du, u); v = np.where(isout, v + dv, v)
        mono = rng.uniform(n_pts) < mono_frac
        ur = np.where(mono, -1.0, ur)
        for j in np.nonzero(vis)[0]:
            e_pose.append(i); e_point.append(j); e_obs.append((u[j], v[j], ur[j])); e_is2.append(inv_sigma2[octave[j]])
            e_out.append(bool(isout[j]))
    pts_init = pts + rng.uniform(3 * n_pts, -perturb[2], perturb[2]).reshape(n_pts, 3)
    return {"poses": poses_init, "pose_flags": flags, "points": pts_init,
            "e_pose": np.array(e_pose, np.int32), "e_point": np.array(e_point, np.int32),
            "e_obs": np.array(e_obs, np.float32), "e_inv_sigma2": np.array(e_is2, np.float32),
            "K": (np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), np.float32(KITTI_BF)),
            "poses_true": poses_true, "points_true": pts, "e_is_outlier": np.array(e_out)}


# ---- projection-matching scenes (SearchByProjection x3) ------------------------------------------------------
def _flip(rng, d, nmax):
    bits = np.unpackbits(d)
    n = int(rng.integers(1, 0, nmax + 1)[0])
    if n:
        bits[rng.integers(n, 0, 256)] ^= 1
    return np.packbits(bits)


def projection_scene(seed, n=2000, m=1500, w=1241, h=376, object_mode=False, th=7.0):
    """A current frame with n keypoints (grid built by the caller) and m source points that project close to some of
    them with similar descriptors.  Returns the 'train' side and both query layouts (frame-to-frame and pre-projected)."""
    rng = Rng(seed)
    fx, fy, cx, cy = KITTI_K
    sig, _ = _level_sigma()
    x = rng.uniform(n, 20, w - 20).astype(np.float32); y = rng.uniform(n, 20, h - 20).astype(np.float32)
    octave = np.searchsorted(np.cumsum(_QUOTAS) / _QUOTAS.sum(), rng.uniform(n)).clip(0, 7).astype(np.int32)
    angle = rng.uniform(n, 0, 360).astype(np.float32)
    z = rng.uniform(n, 4.0, 50.0)
    ur = np.where(rng.uniform(n) < 0.8, x - KITTI_BF / z, -1.0).astype(np.float32)
    desc = rng.integers(n * 32, 0, 256).astype(np.uint8).reshape(n, 32)
    occupied = (rng.uniform(n) < 0.05).astype(np.uint8)
    in_bbox = (rng.uniform(n) < 0.9).astype(np.uint8)
    train = {"x": x, "y": y, "octave": octave, "angle": angle, "u_right": ur, "desc": desc, "occupied": occupied,
             "in_bbox": in_bbox, "grid": (0.0, 0.0, np.float32(64) / np.float32(w), np.float32(48) / np.float32(h))}
    # poses: last at identity, current moved forward + small rotation
    R = _so3_exp(np.radians(rng.uniform(3, -1.0, 1.0)))
    t = np.array([0.05, -0.02, -0.6]) + rng.uniform(3, -0.05, 0.05)
    tcw = np.eye(4, dtype=np.float32); tcw[:3, :3] = R; tcw[:3, 3] = t
    tlw = np.eye(4, dtype=np.float32)
    src = rng.integers(m, 0, n)
    jit = rng.normal(2 * m).reshape(m, 2) * 2.0
    u = x[src] + jit[:, 0]; v = y[src] + jit[:, 1]
    zc = z[src]
    Xc = np.stack([(u - cx) / fx * zc, (v - cy) / fy * zc, zc], 1)
    xw = ((Xc - t) @ R).astype(np.float32)
    q_desc = np.stack([_flip(rng, desc[j], 40) for j in src]) if m else np.zeros((0, 32), np.uint8)
    rand = rng.uniform(m) < 0.15
    q_desc[rand] = rng.integers(int(rand.sum()) * 32, 0, 256).astype(np.uint8).reshape(-1, 32)
    q_oct = np.clip(octave[src] + rng.integers(m, -1, 2), 0, 7).astype(np.int32)
    rot = 12.0 + rng.normal(m) * np.where(rng.uniform(m) < 0.85, 2.0, 70.0)
    q_angle = ((angle[src] + rot) % 360.0).astype(np.float32)
    valid = (rng.uniform(m) < 0.9).astype(np.uint8)
    observed = (rng.uniform(m) < 0.85).astype(np.uint8)
    frame_q = {"valid": valid, "desc": q_desc, "observed": observed, "angle": q_angle, "xw": xw, "octave": q_oct}
    pts_q = {"valid": valid, "desc": q_desc, "observed": observed, "proj_x": u.astype(np.float32), "proj_y": v.astype(np.float32),
             "proj_xr": (u - KITTI_BF / zc).astype(np.float32), "level": q_oct,
             "view_cos": rng.uniform(m, 0.99, 1.0).astype(np.float32)}
    return {"train": train, "frame_query": frame_q, "points_query": pts_q, "tcw": tcw, "tlw": tlw,
            "K6": (np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), np.float32(KITTI_BF), np.float32(KITTI_BF / fx)),
            "bounds": (0.0, float(w), 0.0, float(h)), "scale_factors": sig, "th": th}


def fuse_scene(seed, n=1500, m=800, w=1241, h=376, th=3.0, box=None):
    """A keyframe with n features and m candidate points for ORBmatcher::Fuse: most candidates re-project onto a feature with a
    similar descriptor and a consistent depth; the rest exercise every gate (behind the camera, outside the image / box, outside
    the scale-invariance range, viewing angle above 60 degrees, wrong level, stereo chi-square, far descriptor)."""
    rng = Rng(seed)
    fx, fy, cx, cy = KITTI_K
    sig, inv_sigma2 = _level_sigma()
    x = rng.uniform(n, 20, w - 20).astype(np.float32); y = rng.uniform(n, 20, h - 20).astype(np.float32)
    octave = np.searchsorted(np.cumsum(_QUOTAS) / _QUOTAS.sum(), rng.uniform(n)).clip(0, 7).astype(np.int32)
    z = rng.uniform(n, 4.0, 50.0)
    ur = np.where(rng.uniform(n) < 0.8, x - KITTI_BF / z, -1.0).astype(np.float32)
    desc = rng.integers(n * 32, 0, 256).astype(np.uint8).reshape(n, 32)
    train = {"x": x, "y": y, "octave": octave, "u_right": ur, "desc": desc,
             "grid": (0.0, 0.0, np.float32(64) / np.float32(w), np.float32(48) / np.float32(h))}
    R = _so3_exp(np.radians(rng.uniform(3, -4.0, 4.0))).astype(np.float32)
    t = rng.uniform(3, -1.0, 1.0).astype(np.float32)
    ow = (-(R.T.astype(np.float64) @ t)).astype(np.float32)
    src = rng.integers(m, 0, n)
    jit = rng.normal(2 * m).reshape(m, 2) * 0.7 * sig[octave[src]][:, None]
    zc = z[src] * np.where(rng.uniform(m) < 0.9, 1.0, rng.uniform(m, 0.6, 1.6))          # some with an inconsistent depth
    zc = np.where(rng.uniform(m) < 0.03, -zc, zc)                                          # behind the camera
    u = x[src] + jit[:, 0] + np.where(rng.uniform(m) < 0.04, 3000.0, 0.0)                 # outside the image
    v = y[src] + jit[:, 1]
    Pc = np.stack([(u - cx) / fx * zc, (v - cy) / fy * zc, zc], 1)
    pos = ((Pc - t) @ R).astype(np.float32)
    PO = pos.astype(np.float64) - ow
    dist = np.linalg.norm(PO, axis=1)
    nrm = PO / dist[:, None]
    tilt = rng.uniform(m) < 0.08
    nrm[tilt] = np.roll(nrm[tilt], 1, axis=1) * np.array([1.0, -1.0, 1.0])                 # viewing angle far off
    lvl = np.clip(octave[src] + rng.integers(m, 0, 2) + np.where(rng.uniform(m) < 0.1, 3, 0), 0, 7)
    max_dist = (dist * sig[lvl] * rng.uniform(m, 0.86, 0.99)).astype(np.float32)          # ceil(log(ratio)/log 1.2) == lvl
    far = rng.uniform(m) < 0.05
    max_dist[far] *= np.float32(0.3)                                                      # outside the invariance range
    min_dist = (max_dist / sig[7]).astype(np.float32)
    qd = np.stack([_flip(rng, desc[j], 30) for j in src]) if m else np.zeros((0, 32), np.uint8)
    rand = rng.uniform(m) < 0.1
    qd[rand] = rng.integers(int(rand.sum()) * 32, 0, 256).astype(np.uint8).reshape(-1, 32)
    valid = (rng.uniform(m) < 0.93).astype(np.uint8)
    query = {"valid": valid, "pos": pos, "normal": nrm.astype(np.float32), "min_dist": min_dist, "max_dist": max_dist, "desc": qd}
    bounds = (0.0, float(w), 0.0, float(h)) if box is None else tuple(float(b) for b in box)
    return {"train": train, "query": query, "R": R, "t": t, "ow": ow, "K5": (np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), np.float32(KITTI_BF)),
            "bounds": bounds, "scale_factors": sig, "inv_level_sigma2": inv_sigma2, "log_scale_factor": np.float32(np.log(np.float32(1.2))),
            "n_levels": 8, "th": th, "src": src}


def dynamic_object(seed, n=300, moving=0.0, mono_frac=0.3, outlier_frac=0.05, noise=1.0, valid_frac=0.9):
    """One tracked detection for the reprojection test of Tracking::DynamicStaticDiscrimination: n object points in a car-sized
    cuboid, the object pose in the last frame, the camera poses of both frames and the current observations.  `moving` metres of
    object motion between the frames (0: static object, the chi-squares stay near their noise floor)."""
    rng = Rng(seed)
    fx, fy, cx, cy = KITTI_K
    sig, inv_sigma2 = _level_sigma()
    po = np.stack([rng.uniform(n, -2.0, 2.0), rng.uniform(n, -0.8, 0.8), rng.uniform(n, -0.75, 0.75)], 1)

    def pose7(R, t):
        return np.concatenate([t, _quat_from_R(R)])
    Rco = _Ry(float(rng.uniform(1, -0.6, 0.6)[0])); tco = np.array([float(rng.uniform(1, -4, 4)[0]), 1.0, float(rng.uniform(1, 9, 25)[0])])
    Rl = _so3_exp(np.radians(rng.uniform(3, -2, 2))); tl = rng.uniform(3, -1, 1)
    Rc = _so3_exp(np.radians(rng.uniform(3, -2, 2))) @ Rl; tc = tl + np.array([0.05, 0.0, -0.9]) + rng.uniform(3, -0.05, 0.05)
    # truth: the object moved by `moving` along its own x axis between the frames
    Plc = po @ Rco.T + tco
    Pw = (Plc - tl) @ Rl                                  # world points at the last frame
    Pw_now = Pw + moving * (Rl.T @ Rco)[:, 0]
    Pc = Pw_now @ Rc.T + tc
    octave = np.searchsorted(np.cumsum(_QUOTAS) / _QUOTAS.sum(), rng.uniform(n)).clip(0, 7)
    sd = noise * sig[octave].astype(np.float64)
    u = fx * Pc[:, 0] / Pc[:, 2] + cx + sd * rng.normal(n); v = fy * Pc[:, 1] / Pc[:, 2] + cy + sd * rng.normal(n)
    ur = fx * Pc[:, 0] / Pc[:, 2] + cx - KITTI_BF / Pc[:, 2] + sd * rng.normal(n)
    out = rng.uniform(n) < outlier_frac
    u[out] += rng.uniform(int(out.sum()), 30, 90)
    ur = np.where(rng.uniform(n) < mono_frac, -1.0, np.maximum(ur, 0.0))
    return {"valid": (rng.uniform(n) < valid_frac).astype(np.uint8), "po": po, "obs": np.stack([u, v, ur], 1).astype(np.float32),
            "inv_sigma2": inv_sigma2[octave], "last_tco": pose7(Rco, tco), "last_tcw": pose7(Rl, tl), "cur_tcw": pose7(Rc, tc),
            "K": (fx, fy, cx, cy), "mbf": np.float32(KITTI_BF)}
