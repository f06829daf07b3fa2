"""ctypes wrapper of oracle/liboracle.so — the CPU restatement used ONLY as the checker by tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  Builds the library on first use."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".inc", ".h"))]
    stale = (not os.path.exists(LIB)) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.orc_orb_create.restype = ctypes.c_void_p
        L.orc_orb_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_orb_destroy.argtypes = [ctypes.c_void_p]
        L.orc_orb_tables.argtypes = [ctypes.c_void_p] * 7
        L.orc_orb_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_orb_result.argtypes = [ctypes.c_void_p] * 3
        L.orc_orb_level_dims.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        for f in ("orc_orb_level_padded", "orc_orb_level_blur", "orc_orb_level_cand", "orc_orb_level_kps"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_orb_level_ncand.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_orb_level_nkp.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_fast_atan2.restype = ctypes.c_float
        L.orc_fast_atan2.argtypes = [ctypes.c_float, ctypes.c_float]
        L.orc_distribute.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        L.orc_descriptor_distance.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_hamming_matrix.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_search_bruteforce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_int, ctypes.c_void_p]
        L.orc_pose_optimize.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_float] * 5 + [ctypes.c_void_p] * 4
        L.orc_cfse3_optimize.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_float] * 5 + [ctypes.c_void_p] * 2
        L.orc_object_ba.argtypes = ([ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
                                    + [ctypes.c_void_p] * 4 + [ctypes.c_float] * 5 + [ctypes.c_void_p] * 3)
        L.orc_se3_exp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_se3_log.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_se3_from_mat4f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_se3_to_mat4f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_edge_eval.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_double] * 5 + [ctypes.c_void_p] * 3
        L.orc_huber.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        _lib = L
    return _lib


class OracleORB:
    def __init__(self, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=5):
        self.L = lib()
        self.nlevels = nlevels
        self.h = ctypes.c_void_p(self.L.orc_orb_create(nfeatures, scale, nlevels, ini_th, min_th))

    def __del__(self):
        try:
            self.L.orc_orb_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        f = [np.zeros(n, np.float32) for _ in range(4)]
        q = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, *[a.ctypes.data for a in f], q.ctypes.data, um.ctypes.data)
        return f, q, um

    def run(self, img):
        img = np.ascontiguousarray(img)
        n = self.L.orc_orb_run(self.h, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0])
        self.last_n = n
        kps = np.zeros(max(n, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        self.L.orc_orb_result(self.h, kps.ctypes.data, desc.ctypes.data)
        return kps[:n], desc[:n]

    def run_masked(self, img, mask):
        img = np.ascontiguousarray(img); mask = np.ascontiguousarray(mask)
        self.L.orc_orb_run_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        n = self.L.orc_orb_run_masked(self.h, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], mask.ctypes.data, mask.strides[0])
        self.last_n = n
        kps = np.zeros(max(n, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        self.L.orc_orb_result(self.h, kps.ctypes.data, desc.ctypes.data)
        return kps[:n], desc[:n]

    def level_dims(self, l):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.L.orc_orb_level_dims(self.h, l, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def padded(self, l):
        w, h = self.level_dims(l)
        out = np.zeros((h + 38, w + 38), np.uint8)
        self.L.orc_orb_level_padded(self.h, l, out.ctypes.data)
        return out

    def blur(self, l):
        w, h = self.level_dims(l)
        out = np.zeros((h, w), np.uint8)
        self.L.orc_orb_level_blur(self.h, l, out.ctypes.data)
        return out

    def candidates(self, l):
        n = self.L.orc_orb_level_ncand(self.h, l)
        out = np.zeros((max(n, 1), 3), np.int32)
        self.L.orc_orb_level_cand(self.h, l, out.ctypes.data)
        return out[:n]

    def level_keypoints(self, l):
        n = self.L.orc_orb_level_nkp(self.h, l)
        out = np.zeros(max(n, 1), KEYPOINT_DTYPE)
        self.L.orc_orb_level_kps(self.h, l, out.ctypes.data)
        return out[:n]


class OracleCvORB:
    """the CPU restatement of cv::ORB::create(...)->detectAndCompute(image, mask, ...) (oracle/orb_oracle.cpp: cv_orb_run)"""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, edge=19, fast_th=20):
        self.L = lib()
        self.L.orc_cvorb_create.restype = ctypes.c_void_p
        self.L.orc_cvorb_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.L.orc_cvorb_destroy.argtypes = [ctypes.c_void_p]
        self.L.orc_cvorb_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        self.L.orc_cvorb_result.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.L.orc_cvorb_level_dims.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.L.orc_cvorb_level_plane.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        self.L.orc_cvorb_level_fast.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        self.nlevels = nlevels
        self.h = ctypes.c_void_p(self.L.orc_cvorb_create(nfeatures, scale, nlevels, edge, fast_th))

    def __del__(self):
        try:
            self.L.orc_cvorb_destroy(self.h)
        except Exception:
            pass

    def run(self, img, mask=None):
        img = np.ascontiguousarray(img)
        mp, ms = None, 0
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            mp, ms = mask.ctypes.data, mask.strides[0]
        n = self.L.orc_cvorb_run(self.h, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], mp, ms)
        kps = np.zeros(max(n, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        self.L.orc_cvorb_result(self.h, kps.ctypes.data, desc.ctypes.data)
        return kps[:n], desc[:n]

    def level_dims(self, l):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.L.orc_cvorb_level_dims(self.h, l, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def plane(self, l, what):
        w, h = self.level_dims(l)
        out = np.zeros((h, w), np.uint8)
        self.L.orc_cvorb_level_plane(self.h, l, what, out.ctypes.data)
        return out

    def fast(self, l):
        w, h = self.level_dims(l)
        cap = ((w + 1) // 2) * ((h + 1) // 2)
        out = np.zeros((cap, 4), np.float32)
        n = self.L.orc_cvorb_level_fast(self.h, l, out.ctypes.data, cap)
        return out[:n]


def stereo_match(left, right, mb, mbf):
    """left / right: OracleORB objects after run().  Returns (kept, u_right, depth)."""
    L = lib()
    L.orc_stereo_match.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    n = L.orc_orb_run  # noqa
    nl = left.last_n
    ur = np.zeros(max(nl, 1), np.float32); dp = np.zeros(max(nl, 1), np.float32)
    kept = L.orc_stereo_match(left.h, right.h, mb, mbf, ur.ctypes.data, dp.ctypes.data)
    return kept, ur[:nl], dp[:nl]


def distribute(keys_xyz, minX, maxX, minY, maxY, N):
    keys = np.ascontiguousarray(keys_xyz, np.int32)
    out = np.zeros((N + 16 + 64, 3), np.int32)
    n = lib().orc_distribute(keys.ctypes.data, len(keys), minX, maxX, minY, maxY, N, out.ctypes.data)
    return out[:n]


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(a.ctypes.data, b.ctypes.data)


def hamming_matrix(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros((len(q), len(t)), np.uint16)
    lib().orc_hamming_matrix(q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data)
    return out


def search_bruteforce(p, nnratio, check_ori):
    qd = np.ascontiguousarray(p["q_desc"], np.uint8).reshape(-1, 32)
    td = np.ascontiguousarray(p["t_desc"], np.uint8).reshape(-1, 32)
    qa = np.ascontiguousarray(p["q_angle"], np.float32); ta = np.ascontiguousarray(p["t_angle"], np.float32)
    qv = np.ascontiguousarray(p["q_valid"], np.uint8)
    out = np.full(max(len(td), 1), -1, np.int32)
    n = lib().orc_search_bruteforce(qd.ctypes.data, qa.ctypes.data, qv.ctypes.data, len(qd), td.ctypes.data,
                                    ta.ctypes.data, len(td), nnratio, 1 if check_ori else 0, out.ctypes.data)
    return n, out[:len(td)].copy()


def pose_optimize(p, want_trace=False):
    """p: dict from synth.pose_problem.  Returns (ninliers, tcw float32 4x4, outlier uint8[n], trace)."""
    n = len(p["xw"])
    xw = np.ascontiguousarray(p["xw"], np.float32); obs = np.ascontiguousarray(p["obs"], np.float32)
    is2 = np.ascontiguousarray(p["inv_sigma2"], np.float32); valid = np.ascontiguousarray(p["valid"], np.uint8)
    tcw = np.ascontiguousarray(p["tcw0"], np.float32).copy()
    outlier = np.ascontiguousarray(p.get("outlier0", np.zeros(n, np.uint8)), np.uint8).copy()
    trace = np.zeros((64, 3)); nt = ctypes.c_int(0)
    r = lib().orc_pose_optimize(n, xw.ctypes.data, obs.ctypes.data, is2.ctypes.data, valid.ctypes.data,
                                *[float(k) for k in p["K"]], tcw.ctypes.data, outlier.ctypes.data,
                                trace.ctypes.data, ctypes.byref(nt))
    return r, tcw, outlier, trace[:nt.value]


def cfse3_optimize(objs, K):
    """objs: list of dicts {xo [n,3], obs [n,3], inv_sigma2 [n], valid [n], pose7 [7]}.  Returns (ok, poses7, outliers)."""
    k = len(objs)
    off = np.zeros(k + 1, np.int32)
    for i, o in enumerate(objs):
        off[i + 1] = off[i] + len(o["xo"])
    # (explicit widths: an object that has just left the image has no features, and a 0-row array has no -1 to infer)
    cat = lambda key, dt, wd: np.ascontiguousarray(np.concatenate([np.asarray(o[key], dt).reshape(len(o["xo"]), wd) for o in objs]), dt) if k else np.zeros((0, wd), dt)
    xo, obs, is2, valid = cat("xo", np.float32, 3), cat("obs", np.float32, 3), cat("inv_sigma2", np.float32, 1), cat("valid", np.uint8, 1)
    poses = np.ascontiguousarray(np.stack([o["pose7"] for o in objs]), np.float64).copy() if k else np.zeros((0, 7))
    outlier = np.zeros(max(int(off[-1]), 1), np.uint8)
    r = lib().orc_cfse3_optimize(k, off.ctypes.data, xo.ctypes.data, obs.ctypes.data, is2.ctypes.data, valid.ctypes.data,
                                 *[float(v) for v in K], poses.ctypes.data, outlier.ctypes.data)
    return r, poses, [outlier[off[i]:off[i + 1]].copy() for i in range(k)]


def object_ba(p):
    poses = np.ascontiguousarray(p["poses"], np.float64).copy(); pts = np.ascontiguousarray(p["points"], np.float64).copy()
    flags = np.ascontiguousarray(p["pose_flags"], np.uint8)
    ep = np.ascontiguousarray(p["e_pose"], np.int32); el = np.ascontiguousarray(p["e_point"], np.int32)
    eo = np.ascontiguousarray(p["e_obs"], np.float32); ei = np.ascontiguousarray(p["e_inv_sigma2"], np.float32)
    erase = np.zeros(max(len(ep), 1), np.uint8); trace = np.zeros((64, 3)); nt = ctypes.c_int(0)
    n = lib().orc_object_ba(len(poses), poses.ctypes.data, flags.ctypes.data, len(pts), pts.ctypes.data, len(ep),
                            ep.ctypes.data, el.ctypes.data, eo.ctypes.data, ei.ctypes.data, *[float(v) for v in p["K"]],
                            erase.ctypes.data, trace.ctypes.data, ctypes.byref(nt))
    return n, poses, pts, erase[:len(ep)].copy(), trace[:nt.value]


def se3_exp(u, norollpitch=False):
    u = np.ascontiguousarray(u, np.float64); out = np.zeros(7)
    lib().orc_se3_exp(u.ctypes.data, 1 if norollpitch else 0, out.ctypes.data)
    return out


def se3_log(p7):
    p7 = np.ascontiguousarray(p7, np.float64); out = np.zeros(6)
    lib().orc_se3_log(p7.ctypes.data, out.ctypes.data)
    return out


def se3_from_mat4f(m):
    m = np.ascontiguousarray(m, np.float32); out = np.zeros(7)
    lib().orc_se3_from_mat4f(m.ctypes.data, out.ctypes.data)
    return out


def se3_to_mat4f(p7):
    p7 = np.ascontiguousarray(p7, np.float64); out = np.zeros((4, 4), np.float32)
    lib().orc_se3_to_mat4f(p7.ctypes.data, out.ctypes.data)
    return out


def edge_eval(etype, pose7, X, obs, K):
    pose7 = np.ascontiguousarray(pose7, np.float64); X = np.ascontiguousarray(X, np.float64); obs = np.ascontiguousarray(obs, np.float64)
    err = np.zeros(3); Jp = np.zeros((3, 6)); Jx = np.zeros((3, 3))
    lib().orc_edge_eval(etype, pose7.ctypes.data, X.ctypes.data, obs.ctypes.data, *[float(v) for v in K], err.ctypes.data,
                        Jp.ctypes.data, Jx.ctypes.data)
    return err, Jp, Jx


def huber(e, delta):
    out = np.zeros(3)
    lib().orc_huber(float(e), float(delta), out.ctypes.data)
    return out


class _OrcTrain(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("octave", ctypes.c_void_p),
                ("angle", ctypes.c_void_p), ("u_right", ctypes.c_void_p), ("desc", ctypes.c_void_p),
                ("occupied", ctypes.c_void_p), ("in_bbox", ctypes.c_void_p), ("cell_off", ctypes.c_void_p),
                ("cell_idx", ctypes.c_void_p), ("min_x", ctypes.c_float), ("min_y", ctypes.c_float),
                ("gw_inv", ctypes.c_float), ("gh_inv", ctypes.c_float)]


def _orc_train(F, keep):
    n = len(F["x"])
    a = [np.ascontiguousarray(F["x"], np.float32), np.ascontiguousarray(F["y"], np.float32), np.ascontiguousarray(F["octave"], np.int32),
         np.ascontiguousarray(F["angle"], np.float32), np.ascontiguousarray(F["u_right"], np.float32),
         np.ascontiguousarray(F["desc"], np.uint8), np.ascontiguousarray(F["occupied"], np.uint8),
         np.ascontiguousarray(F.get("in_bbox", np.ones(n, np.uint8)), np.uint8),
         np.ascontiguousarray(F["cell_off"], np.int32), np.ascontiguousarray(F["cell_idx"], np.int32)]
    keep.append(a)
    g = [float(v) for v in F["grid"]]
    return _OrcTrain(n, *[x.ctypes.data for x in a], *g)


def search_projection_frame(pr, check_ori=True):
    keep = []
    T = _orc_train(pr["train"], keep)
    q = pr["query"]
    m = len(q["valid"])
    c = lambda k, dt: np.ascontiguousarray(q[k], dt)
    xw, valid, oc, ang, desc, obs = c("xw", np.float32), c("valid", np.uint8), c("octave", np.int32), c("angle", np.float32), c("desc", np.uint8), c("observed", np.uint8)
    tcw = np.ascontiguousarray(pr["tcw"], np.float32); tlw = np.ascontiguousarray(pr["tlw"], np.float32)
    K6 = np.ascontiguousarray(pr["K6"], np.float32); b4 = np.ascontiguousarray(pr["bounds"], np.float32)
    sf = np.ascontiguousarray(pr["scale_factors"], np.float32)
    out = np.full(max(T.n, 1), -1, np.int32)
    f = lib().orc_search_projection_frame
    f.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 11 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    n = f(ctypes.byref(T), m, xw.ctypes.data, valid.ctypes.data, oc.ctypes.data, ang.ctypes.data, desc.ctypes.data, obs.ctypes.data,
          tcw.ctypes.data, tlw.ctypes.data, K6.ctypes.data, b4.ctypes.data, sf.ctypes.data, float(pr["th"]),
          1 if pr.get("mono") else 0, 1 if check_ori else 0, out.ctypes.data)
    return n, out[:T.n].copy()


def search_projection_points(pr, nnratio):
    keep = []
    T = _orc_train(pr["train"], keep)
    q = pr["query"]
    m = len(q["valid"])
    c = lambda k, dt: np.ascontiguousarray(q[k], dt)
    valid, px, py, pxr, lvl, vc, desc, obs = (c("valid", np.uint8), c("proj_x", np.float32), c("proj_y", np.float32), c("proj_xr", np.float32),
                                              c("level", np.int32), c("view_cos", np.float32), c("desc", np.uint8), c("observed", np.uint8))
    sf = np.ascontiguousarray(pr["scale_factors"], np.float32)
    out = np.full(max(T.n, 1), -1, np.int32)
    f = lib().orc_search_projection_points
    f.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    n = f(ctypes.byref(T), m, valid.ctypes.data, px.ctypes.data, py.ctypes.data, pxr.ctypes.data, lvl.ctypes.data, vc.ctypes.data,
          desc.ctypes.data, obs.ctypes.data, sf.ctypes.data, float(pr["th"]), float(nnratio), 1 if pr.get("object") else 0, out.ctypes.data)
    return n, out[:T.n].copy()


def distinctive_descriptors(desc_lists):
    n = len(desc_lists)
    off = np.zeros(n + 1, np.int32)
    for i, d in enumerate(desc_lists):
        off[i + 1] = off[i] + len(d)
    cat = np.ascontiguousarray(np.concatenate([np.asarray(d, np.uint8).reshape(-1, 32) for d in desc_lists] + [np.zeros((1, 32), np.uint8)]))
    best = np.zeros(n, np.int32)
    f = lib().orc_distinctive_descriptors
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    f(cat.ctypes.data, off.ctypes.data, n, best.ctypes.data)
    return best


def stereo_match_keys(left, right, kps_l, desc_l, kps_r, desc_r, mb, mbf):
    """Frame::ComputeObjStereoMatches on caller key sets; left / right: OracleORB objects after run()."""
    L = lib()
    f = L.orc_stereo_match_keys
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    kl = np.ascontiguousarray(kps_l, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kps_r, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(desc_l, np.uint8).reshape(-1, 32); dr = np.ascontiguousarray(desc_r, np.uint8).reshape(-1, 32)
    ur = np.full(max(len(kl), 1), -1.0, np.float32); dp = np.full(max(len(kl), 1), -1.0, np.float32)
    kept = f(left.h, right.h, kl.ctypes.data, dl.ctypes.data, len(kl), kr.ctypes.data, dr.ctypes.data, len(kr), mb, mbf, ur.ctypes.data, dp.ctypes.data)
    return kept, ur[:len(kl)], dp[:len(kl)]


def fuse_search(pr):
    keep = []
    F = dict(pr["train"])
    nt = len(F["x"])
    F.setdefault("angle", np.zeros(nt, np.float32)); F.setdefault("occupied", np.zeros(nt, np.uint8))
    T = _orc_train(F, keep)
    q = pr["query"]
    m = len(q["valid"])
    c = lambda k, dt: np.ascontiguousarray(q[k], dt)
    valid, pos, nor, mind, maxd, desc = c("valid", np.uint8), c("pos", np.float32), c("normal", np.float32), c("min_dist", np.float32), c("max_dist", np.float32), c("desc", np.uint8)
    f32 = lambda v: np.ascontiguousarray(v, np.float32)
    R, t, ow, K5, sf, is2 = f32(pr["R"]), f32(pr["t"]), f32(pr["ow"]), f32(pr["K5"]), f32(pr["scale_factors"]), f32(pr["inv_level_sigma2"])
    b = np.ascontiguousarray(pr["bounds"], np.float64)
    bi = np.full(max(m, 1), -1, np.int32); bd = np.full(max(m, 1), 256, np.int32)
    f = lib().orc_fuse_search
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 13 + [ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    f(ctypes.byref(T), m, valid.ctypes.data, pos.ctypes.data, nor.ctypes.data, mind.ctypes.data, maxd.ctypes.data, desc.ctypes.data,
      R.ctypes.data, t.ctypes.data, ow.ctypes.data, K5.ctypes.data, b.ctypes.data, sf.ctypes.data, is2.ctypes.data,
      float(np.float32(pr["log_scale_factor"])), int(pr["n_levels"]), float(np.float32(pr["th"])), bi.ctypes.data, bd.ctypes.data)
    return bi[:m].copy(), bd[:m].copy()


def dynamic_discrimination(o):
    valid = np.ascontiguousarray(o["valid"], np.uint8); po = np.ascontiguousarray(o["po"], np.float64)
    obs = np.ascontiguousarray(o["obs"], np.float32); is2 = np.ascontiguousarray(o["inv_sigma2"], np.float32)
    p7 = [np.ascontiguousarray(o[k], np.float64) for k in ("last_tco", "last_tcw", "cur_tcw")]
    avg = np.zeros(2); cnt = np.zeros(2, np.int32)
    f = lib().orc_dynamic_discrimination
    f.restype = None
    f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_double] * 4 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    f(len(valid), valid.ctypes.data, po.ctypes.data, obs.ctypes.data, is2.ctypes.data, p7[0].ctypes.data, p7[1].ctypes.data, p7[2].ctypes.data,
      *[float(k) for k in o["K"]], float(np.float32(o["mbf"])), avg.ctypes.data, cnt.ctypes.data)
    return avg[0], avg[1], int(cnt[0]), int(cnt[1])
