"""Host-side mirror of the hot static members of ORB_SLAM2::Optimizer (/root/reference/include/Optimizer.h:51-61)
on top of the C-ABI.  The reference reads its inputs out of Frame / MapObject / ObjectKeyFrame objects; here the
same quantities are passed as arrays (see include/pointslot_hip.h for the field-by-field mapping)."""
import ctypes

import numpy as np

from ._lib import lib, check


class _PoseProblem(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("xw", ctypes.c_void_p), ("obs", ctypes.c_void_p), ("inv_sigma2", ctypes.c_void_p),
                ("valid", ctypes.c_void_p), ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float),
                ("cy", ctypes.c_float), ("bf", ctypes.c_float), ("tcw", ctypes.c_float * 16), ("outlier", ctypes.c_void_p),
                ("result", ctypes.c_int32)]


class _Cfse3Problem(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("off", ctypes.c_void_p), ("xo", ctypes.c_void_p), ("obs", ctypes.c_void_p),
                ("inv_sigma2", ctypes.c_void_p), ("valid", ctypes.c_void_p), ("fx", ctypes.c_float), ("fy", ctypes.c_float),
                ("cx", ctypes.c_float), ("cy", ctypes.c_float), ("bf", ctypes.c_float), ("poses7", ctypes.c_void_p),
                ("outlier", ctypes.c_void_p), ("result", ctypes.c_int32)]


class _BaProblem(ctypes.Structure):
    _fields_ = [("np", ctypes.c_int32), ("nl", ctypes.c_int32), ("ne", ctypes.c_int32), ("poses7", ctypes.c_void_p),
                ("pose_flags", ctypes.c_void_p), ("points", ctypes.c_void_p), ("e_pose", ctypes.c_void_p),
                ("e_point", ctypes.c_void_p), ("e_obs", ctypes.c_void_p), ("e_inv_sigma2", ctypes.c_void_p),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
                ("bf", ctypes.c_float), ("erase", ctypes.c_void_p), ("n_erased", ctypes.c_int32),
                ("iterations", ctypes.c_int32), ("trials", ctypes.c_int32), ("n_trace", ctypes.c_int32),
                ("trace", ctypes.c_void_p)]


lib.ps_optimizer_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
lib.ps_optimizer_destroy.argtypes = [ctypes.c_void_p]
lib.ps_optimizer_destroy.restype = None
lib.ps_optimizer_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.ps_optimizer_enable_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ps_optimizer_get_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.ps_pose_optimize_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_PoseProblem), ctypes.c_int]
lib.ps_cfse3_optimize_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Cfse3Problem), ctypes.c_int]
lib.ps_object_ba_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_BaProblem), ctypes.c_int]
lib.ps_se3_from_mat4f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.ps_se3_to_mat4f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]


class _DynProblem(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("valid", ctypes.c_void_p), ("po", ctypes.c_void_p), ("obs", ctypes.c_void_p),
                ("inv_sigma2", ctypes.c_void_p), ("last_tco", ctypes.c_double * 7), ("last_tcw", ctypes.c_double * 7),
                ("cur_tcw", ctypes.c_double * 7), ("fx", ctypes.c_double), ("fy", ctypes.c_double), ("cx", ctypes.c_double),
                ("cy", ctypes.c_double), ("mbf", ctypes.c_float), ("mono_avg", ctypes.c_double), ("stereo_avg", ctypes.c_double),
                ("mono_n", ctypes.c_int32), ("stereo_n", ctypes.c_int32)]


lib.ps_dynamic_discrimination_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_DynProblem), ctypes.c_int]


def se3_from_mat4f(m):
    m = np.ascontiguousarray(m, np.float32); out = np.zeros(7)
    check(lib.ps_se3_from_mat4f(m.ctypes.data, out.ctypes.data))
    return out


def se3_to_mat4f(p7):
    p7 = np.ascontiguousarray(p7, np.float64); out = np.zeros((4, 4), np.float32)
    check(lib.ps_se3_to_mat4f(p7.ctypes.data, out.ctypes.data))
    return out


class Optimizer:
    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        check(lib.ps_optimizer_create(device, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib.ps_optimizer_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_kernel_ms(self):
        ms = ctypes.c_float(0)
        check(lib.ps_optimizer_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def enable_trace(self, on=True):
        check(lib.ps_optimizer_enable_trace(self._h, 1 if on else 0))

    def get_trace(self, problem):
        out = np.zeros((64, 3)); n = ctypes.c_int(0)
        check(lib.ps_optimizer_get_trace(self._h, problem, out.ctypes.data, 64, ctypes.byref(n)))
        return out[:n.value].copy()

    def PoseOptimization(self, frames):
        """frames: list of dicts {xw [n,3] f32, obs [n,3] f32 (u,v,uR), inv_sigma2 [n] f32, valid [n] u8,
        K (fx,fy,cx,cy,bf), tcw0 4x4 f32, optional outlier0 [n] u8}.  Returns a list of
        (ninliers, tcw float32 4x4, outlier uint8[n]) — PoseOptimization's return value, SetPose argument, mvbOutlier."""
        n = len(frames)
        arr = (_PoseProblem * n)()
        keep = []
        for i, f in enumerate(frames):
            xw = np.ascontiguousarray(f["xw"], np.float32); obs = np.ascontiguousarray(f["obs"], np.float32)
            is2 = np.ascontiguousarray(f["inv_sigma2"], np.float32); valid = np.ascontiguousarray(f["valid"], np.uint8)
            outl = np.ascontiguousarray(f.get("outlier0", np.zeros(len(xw), np.uint8)), np.uint8).copy()
            keep.append((xw, obs, is2, valid, outl))
            K = [float(v) for v in f["K"]]
            p = arr[i]
            p.n = len(xw); p.xw = xw.ctypes.data; p.obs = obs.ctypes.data; p.inv_sigma2 = is2.ctypes.data
            p.valid = valid.ctypes.data; p.fx, p.fy, p.cx, p.cy, p.bf = K
            p.tcw = (ctypes.c_float * 16)(*np.asarray(f["tcw0"], np.float32).reshape(16))
            p.outlier = outl.ctypes.data
        check(lib.ps_pose_optimize_batch(self._h, arr, n))
        return [(arr[i].result, np.array(arr[i].tcw, np.float32).reshape(4, 4), keep[i][4]) for i in range(n)]

    def CFSE3ObjStateOptimization(self, frames):
        """frames: list of dicts {objs: [ {xo, obs, inv_sigma2, valid, pose7} ... ], K}.  Returns a list of
        (ok, poses7 [k,7], [outlier arrays])."""
        n = len(frames)
        arr = (_Cfse3Problem * n)()
        keep = []
        for i, f in enumerate(frames):
            objs = f["objs"]; k = len(objs)
            off = np.zeros(k + 1, np.int32)
            for j, o in enumerate(objs):
                off[j + 1] = off[j] + len(o["xo"])
            def cat(key, dt, w):
                if k == 0:
                    return np.zeros((1, w), dt)
                return np.ascontiguousarray(np.concatenate([np.asarray(o[key], dt).reshape(len(o["xo"]), w) for o in objs] + [np.zeros((1, w), dt)]))
            xo, obs, is2, valid = cat("xo", np.float32, 3), cat("obs", np.float32, 3), cat("inv_sigma2", np.float32, 1), cat("valid", np.uint8, 1)
            poses = np.ascontiguousarray(np.stack([o["pose7"] for o in objs]), np.float64).copy() if k else np.zeros((1, 7))
            outl = np.zeros(int(off[-1]) + 1, np.uint8)
            keep.append((off, xo, obs, is2, valid, poses, outl))
            K = [float(v) for v in f["K"]]
            p = arr[i]
            p.k = k; p.off = off.ctypes.data; p.xo = xo.ctypes.data; p.obs = obs.ctypes.data; p.inv_sigma2 = is2.ctypes.data
            p.valid = valid.ctypes.data; p.fx, p.fy, p.cx, p.cy, p.bf = K
            p.poses7 = poses.ctypes.data; p.outlier = outl.ctypes.data
        check(lib.ps_cfse3_optimize_batch(self._h, arr, n))
        out = []
        for i, f in enumerate(frames):
            off, _, _, _, _, poses, outl = keep[i]
            k = len(f["objs"])
            out.append((arr[i].result, poses[:k].copy(), [outl[off[j]:off[j + 1]].copy() for j in range(k)]))
        return out

    def ObjectLocalBundleAdjustment(self, graphs):
        """graphs: list of dicts {poses [np,7], pose_flags [np], points [nl,3], e_pose, e_point, e_obs [ne,3],
        e_inv_sigma2 [ne], K} (one collected graph per object).  Returns a list of dicts
        {poses, points, erase, n_erased, iterations, trials, trace}."""
        n = len(graphs)
        arr = (_BaProblem * n)()
        keep = []
        for i, g in enumerate(graphs):
            poses = np.ascontiguousarray(g["poses"], np.float64).copy(); pts = np.ascontiguousarray(g["points"], np.float64).copy()
            flags = np.ascontiguousarray(g["pose_flags"], np.uint8)
            ep = np.ascontiguousarray(g["e_pose"], np.int32); el = np.ascontiguousarray(g["e_point"], np.int32)
            eo = np.ascontiguousarray(g["e_obs"], np.float32); ei = np.ascontiguousarray(g["e_inv_sigma2"], np.float32)
            erase = np.zeros(max(len(ep), 1), np.uint8); trace = np.zeros((40, 3))
            keep.append((poses, pts, flags, ep, el, eo, ei, erase, trace))
            K = [float(v) for v in g["K"]]
            p = arr[i]
            p.np, p.nl, p.ne = len(poses), len(pts), len(ep)
            p.poses7 = poses.ctypes.data; p.pose_flags = flags.ctypes.data; p.points = pts.ctypes.data
            p.e_pose = ep.ctypes.data; p.e_point = el.ctypes.data; p.e_obs = eo.ctypes.data; p.e_inv_sigma2 = ei.ctypes.data
            p.fx, p.fy, p.cx, p.cy, p.bf = K
            p.erase = erase.ctypes.data; p.trace = trace.ctypes.data
        check(lib.ps_object_ba_batch(self._h, arr, n))
        out = []
        for i in range(n):
            poses, pts, _, ep, _, _, _, erase, trace = keep[i]
            out.append({"poses": poses, "points": pts, "erase": erase[:len(ep)].copy(), "n_erased": arr[i].n_erased,
                        "iterations": arr[i].iterations, "trials": arr[i].trials, "trace": trace[:arr[i].n_trace].copy()})
        return out


def _dynamic_static_discrimination(self, objects):
    """Reprojection test of Tracking::DynamicStaticDiscrimination.  objects: list of dicts {valid [n] u8, po [n,3] f64, obs [n,3] f32
    (x, y, uR), inv_sigma2 [n] f32, last_tco, last_tcw, cur_tcw (7 doubles each), K (fx, fy, cx, cy), mbf}.
    Returns a list of (mono_avg, stereo_avg, mono_n, stereo_n)."""
    n = len(objects)
    arr = (_DynProblem * n)()
    keep = []
    for i, o in enumerate(objects):
        a = (np.ascontiguousarray(o["valid"], np.uint8), np.ascontiguousarray(o["po"], np.float64).reshape(-1, 3),
             np.ascontiguousarray(o["obs"], np.float32).reshape(-1, 3), np.ascontiguousarray(o["inv_sigma2"], np.float32))
        keep.append(a)
        p = arr[i]
        p.n = len(a[0]); p.valid, p.po, p.obs, p.inv_sigma2 = [x.ctypes.data for x in a]
        p.last_tco = (ctypes.c_double * 7)(*np.asarray(o["last_tco"], np.float64)); p.last_tcw = (ctypes.c_double * 7)(*np.asarray(o["last_tcw"], np.float64))
        p.cur_tcw = (ctypes.c_double * 7)(*np.asarray(o["cur_tcw"], np.float64))
        p.fx, p.fy, p.cx, p.cy = [float(v) for v in o["K"]]
        p.mbf = float(o["mbf"])
    check(lib.ps_dynamic_discrimination_batch(self._h, arr, n))
    return [(arr[i].mono_avg, arr[i].stereo_avg, arr[i].mono_n, arr[i].stereo_n) for i in range(n)]


Optimizer.DynamicStaticDiscrimination = _dynamic_static_discrimination
