"""Host-side mirror of the hot members of ORB_SLAM2::ORBmatcher (/root/reference/include/ORBmatcher.h:47-118)
on top of the C-ABI.  The reference writes MapPoint pointers into the Frame; here the same assignments come
back as index arrays (index of the source point, -1 for NULL)."""
import ctypes

import numpy as np

from ._lib import lib, check


class _BfProblem(ctypes.Structure):
    _fields_ = [("q_desc", ctypes.c_void_p), ("q_angle", ctypes.c_void_p), ("q_valid", ctypes.c_void_p),
                ("nq", ctypes.c_int32), ("t_desc", ctypes.c_void_p), ("t_angle", ctypes.c_void_p),
                ("nt", ctypes.c_int32), ("query_of_train", ctypes.c_void_p), ("nmatches", ctypes.c_int32)]


lib.ps_matcher_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
lib.ps_matcher_destroy.argtypes = [ctypes.c_void_p]
lib.ps_matcher_destroy.restype = None
lib.ps_hamming_matrix.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                  ctypes.c_void_p]
lib.ps_match_bruteforce.argtypes = [ctypes.c_void_p, ctypes.POINTER(_BfProblem), ctypes.c_int, ctypes.c_float,
                                    ctypes.c_int]


lib.ps_matcher_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.ps_distinctive_descriptors.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]


class ORBmatcher:
    TH_HIGH = 100
    TH_LOW = 50
    TH_HIGH_FORDYNAMIC = 130
    RADIUS_FORDYNAMIC = 5
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._h = ctypes.c_void_p()
        check(lib.ps_matcher_create(device, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib.ps_matcher_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_kernel_ms(self):
        v = ctypes.c_float(0)
        check(lib.ps_matcher_last_kernel_ms(self._h, ctypes.byref(v)))
        return v.value

    def DescriptorDistanceMatrix(self, q, t):
        """bulk ORBmatcher::DescriptorDistance: uint16 [nq, nt]"""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), len(t)), np.uint16)
        check(lib.ps_hamming_matrix(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data))
        return out

    def ComputeDistinctiveDescriptors(self, desc_lists):
        """desc_lists: one uint8 [n_i, 32] array per map point (its observations).  Returns int32 best index per point."""
        n = len(desc_lists)
        off = np.zeros(n + 1, np.int32)
        for i, d in enumerate(desc_lists):
            off[i + 1] = off[i] + len(d)
        cat = np.ascontiguousarray(np.concatenate([np.asarray(d, np.uint8).reshape(-1, 32) for d in desc_lists] + [np.zeros((1, 32), np.uint8)]))
        best = np.zeros(n, np.int32)
        check(lib.ps_distinctive_descriptors(self._h, cat.ctypes.data, off.ctypes.data, n, best.ctypes.data))
        return best

    def SearchByBruceMatching(self, problems):
        """problems: list of dicts {q_desc [nq,32], q_angle [nq], q_valid [nq], t_desc [nt,32], t_angle [nt]}
        (one per tracked object).  Returns a list of (nmatches, query_of_train int32[nt])."""
        n = len(problems)
        arr = (_BfProblem * n)()
        keep = []
        for i, p in enumerate(problems):
            qd = np.ascontiguousarray(p["q_desc"], np.uint8).reshape(-1, 32)
            td = np.ascontiguousarray(p["t_desc"], np.uint8).reshape(-1, 32)
            qa = np.ascontiguousarray(p["q_angle"], np.float32)
            ta = np.ascontiguousarray(p["t_angle"], np.float32)
            qv = np.ascontiguousarray(p["q_valid"], np.uint8)
            out = np.full(max(len(td), 1), -1, np.int32)
            keep.append((qd, td, qa, ta, qv, out))
            arr[i] = _BfProblem(qd.ctypes.data, qa.ctypes.data, qv.ctypes.data, len(qd), td.ctypes.data,
                                ta.ctypes.data, len(td), out.ctypes.data, 0)
        check(lib.ps_match_bruteforce(self._h, arr, n, self.mfNNratio, 1 if self.mbCheckOrientation else 0))
        return [(arr[i].nmatches, keep[i][5][:len(keep[i][1])].copy()) for i in range(n)]


# ---- windowed matching (the three SearchByProjection overloads) ---------------------------------------------
FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48      # /root/reference/include/Frame.h:40-41


def build_grid(x, y, min_x, min_y, gw_inv, gh_inv):
    """Frame::AssignFeaturesToGrid / PosInGrid (/root/reference/src/Frame.cc:1636-1656, 2027-2037) as CSR:
    returns (cell_off int32[64*48+1], cell_idx int32[<=n]) with cell = ix * 48 + iy, insertion order inside a cell."""
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    fx = (x - np.float32(min_x)) * np.float32(gw_inv); fy = (y - np.float32(min_y)) * np.float32(gh_inv)
    px = np.where(fx >= 0, np.floor(fx + np.float32(0.5)), np.ceil(fx - np.float32(0.5))).astype(np.int64)   # C round()
    py = np.where(fy >= 0, np.floor(fy + np.float32(0.5)), np.ceil(fy - np.float32(0.5))).astype(np.int64)
    ok = (px >= 0) & (px < FRAME_GRID_COLS) & (py >= 0) & (py < FRAME_GRID_ROWS)
    cell = px * FRAME_GRID_ROWS + py
    idx = np.nonzero(ok)[0]
    order = np.argsort(cell[idx], kind="stable")
    cell_idx = idx[order].astype(np.int32)
    counts = np.bincount(cell[idx], minlength=FRAME_GRID_COLS * FRAME_GRID_ROWS)
    cell_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return cell_off, cell_idx


class _ProjTrain(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("octave", ctypes.c_void_p),
                ("angle", ctypes.c_void_p), ("u_right", ctypes.c_void_p), ("desc", ctypes.c_void_p),
                ("occupied", ctypes.c_void_p), ("in_bbox", ctypes.c_void_p), ("cell_off", ctypes.c_void_p),
                ("cell_idx", ctypes.c_void_p), ("min_x", ctypes.c_float), ("min_y", ctypes.c_float),
                ("grid_w_inv", ctypes.c_float), ("grid_h_inv", ctypes.c_float)]


class _ProjProblem(ctypes.Structure):
    _fields_ = [("train", _ProjTrain), ("nq", ctypes.c_int32), ("q_valid", ctypes.c_void_p), ("q_desc", ctypes.c_void_p),
                ("q_observed", ctypes.c_void_p), ("q_angle", ctypes.c_void_p), ("q_u", ctypes.c_void_p),
                ("q_v", ctypes.c_void_p), ("q_ur", ctypes.c_void_p), ("q_radius", ctypes.c_void_p),
                ("q_radius_er", ctypes.c_void_p), ("q_min_level", ctypes.c_void_p), ("q_max_level", ctypes.c_void_p),
                ("q_xw", ctypes.c_void_p), ("q_octave", ctypes.c_void_p), ("frame_mode", ctypes.c_int32),
                ("mono", ctypes.c_int32), ("tcw", ctypes.c_float * 16), ("tlw", ctypes.c_float * 16),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
                ("mbf", ctypes.c_float), ("mb", ctypes.c_float), ("bounds", ctypes.c_float * 4),
                ("scale_factors", ctypes.c_float * 8), ("th", ctypes.c_float), ("th_dist", ctypes.c_int32),
                ("ratio_test", ctypes.c_int32), ("nn_ratio", ctypes.c_float), ("check_orientation", ctypes.c_int32),
                ("use_bbox", ctypes.c_int32), ("match_of_train", ctypes.c_void_p), ("nmatches", ctypes.c_int32)]


lib.ps_search_by_projection.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ProjProblem), ctypes.c_int]


def _arr(a, dt):
    return np.ascontiguousarray(a, dt)


def _fill_train(p, F, keep):
    n = len(F["x"])
    a = dict(x=_arr(F["x"], np.float32), y=_arr(F["y"], np.float32), octave=_arr(F["octave"], np.int32),
             angle=_arr(F["angle"], np.float32), u_right=_arr(F["u_right"], np.float32),
             desc=_arr(F["desc"], np.uint8).reshape(-1, 32), occupied=_arr(F["occupied"], np.uint8),
             in_bbox=_arr(F.get("in_bbox", np.ones(n, np.uint8)), np.uint8),
             cell_off=_arr(F["cell_off"], np.int32), cell_idx=_arr(F["cell_idx"], np.int32))
    keep.append(a)
    t = p.train
    t.n = n
    for k, v in a.items():
        setattr(t, k, v.ctypes.data)
    t.min_x, t.min_y, t.grid_w_inv, t.grid_h_inv = [float(v) for v in F["grid"]]
    return n


def _search_by_projection(self, problems, partial_ok=False):
    """problems: list of dicts, see SearchByProjectionFrame / SearchByProjectionPoints for the two layouts.  partial_ok: when a
    problem overflows even the wide candidate store the call fails with PS_ERR_CAPACITY after serving all the others; with
    partial_ok the results come back anyway, nmatches = -1 marking the problems that failed."""
    n = len(problems)
    arr = (_ProjProblem * n)()
    keep, outs = [], []
    for i, pr in enumerate(problems):
        p = arr[i]
        nt = _fill_train(p, pr["train"], keep)
        q = pr["query"]
        nq = len(q["valid"])
        a = dict(q_valid=_arr(q["valid"], np.uint8), q_desc=_arr(q["desc"], np.uint8).reshape(-1, 32),
                 q_observed=_arr(q["observed"], np.uint8))
        p.frame_mode = 1 if pr["mode"] == "frame" else 0
        if p.frame_mode:
            a.update(q_angle=_arr(q["angle"], np.float32), q_xw=_arr(q["xw"], np.float32), q_octave=_arr(q["octave"], np.int32))
            p.tcw = (ctypes.c_float * 16)(*np.asarray(pr["tcw"], np.float32).reshape(16))
            p.tlw = (ctypes.c_float * 16)(*np.asarray(pr["tlw"], np.float32).reshape(16))
            p.fx, p.fy, p.cx, p.cy, p.mbf, p.mb = [float(v) for v in pr["K6"]]
            p.bounds = (ctypes.c_float * 4)(*[float(v) for v in pr["bounds"]])
            p.th = float(pr["th"]); p.mono = 1 if pr.get("mono") else 0
            p.th_dist = ORBmatcher.TH_HIGH; p.ratio_test = 0; p.nn_ratio = self.mfNNratio
            p.check_orientation = 1 if self.mbCheckOrientation else 0; p.use_bbox = 0
        else:
            obj = bool(pr.get("object"))
            th = np.float32(pr["th"])
            lvl = np.asarray(q["level"], np.int32)
            sf = np.asarray(pr["scale_factors"], np.float32)
            r = np.where(np.asarray(q["view_cos"], np.float32) > np.float32(0.998), np.float32(2.5), np.float32(4.0)).astype(np.float32)
            if float(th) != 1.0:
                r = (r * th).astype(np.float32)
            rer = (r * sf[lvl]).astype(np.float32)
            a.update(q_u=_arr(q["proj_x"], np.float32), q_v=_arr(q["proj_y"], np.float32), q_ur=_arr(q["proj_xr"], np.float32),
                     q_radius=np.full(nq, 5.0, np.float32) if obj else rer, q_radius_er=rer,
                     q_min_level=(lvl - 1).astype(np.int32), q_max_level=(lvl + 1 if obj else lvl).astype(np.int32))
            p.th_dist = ORBmatcher.TH_HIGH_FORDYNAMIC if obj else ORBmatcher.TH_HIGH
            p.ratio_test = 1; p.nn_ratio = self.mfNNratio; p.check_orientation = 0; p.use_bbox = 1 if obj else 0
        p.scale_factors = (ctypes.c_float * 8)(*np.asarray(pr["scale_factors"], np.float32))
        p.nq = nq
        for k, v in a.items():
            setattr(p, k, v.ctypes.data)
        out = np.full(max(nt, 1), -1, np.int32)
        p.match_of_train = out.ctypes.data
        keep.append(a); outs.append((out, nt))
    rc = lib.ps_search_by_projection(self._h, arr, n)
    if not (partial_ok and rc == -3):
        check(rc)
    return [(arr[i].nmatches, outs[i][0][:outs[i][1]].copy()) for i in range(n)]


ORBmatcher.SearchByProjection = _search_by_projection


class _FuseProblem(ctypes.Structure):
    _fields_ = [("train", _ProjTrain), ("nq", ctypes.c_int32), ("q_valid", ctypes.c_void_p), ("q_pos", ctypes.c_void_p),
                ("q_normal", ctypes.c_void_p), ("q_min_dist", ctypes.c_void_p), ("q_max_dist", ctypes.c_void_p), ("q_desc", ctypes.c_void_p),
                ("rcw", ctypes.c_float * 9), ("tcw", ctypes.c_float * 3), ("ow", ctypes.c_float * 3),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float), ("bf", ctypes.c_float),
                ("bounds", ctypes.c_double * 4), ("scale_factors", ctypes.c_float * 8), ("inv_level_sigma2", ctypes.c_float * 8),
                ("log_scale_factor", ctypes.c_float), ("n_levels", ctypes.c_int32), ("th", ctypes.c_float),
                ("best_idx", ctypes.c_void_p), ("best_dist", ctypes.c_void_p)]


lib.ps_fuse_search.argtypes = [ctypes.c_void_p, ctypes.POINTER(_FuseProblem), ctypes.c_int]


def _fuse_search(self, problems):
    """The search half of ORBmatcher::Fuse (KeyFrame / ObjectKeyFrame variants).  problems: list of dicts
    {train (keyframe features + grid, as for SearchByProjection), query {valid, pos [m,3], normal [m,3], min_dist, max_dist, desc},
     R 3x3, t 3, ow 3, K5 (fx, fy, cx, cy, bf), bounds (minX, maxX, minY, maxY), scale_factors, inv_level_sigma2,
     log_scale_factor, n_levels, th}.  Returns a list of (best_idx int32[m], best_dist int32[m])."""
    n = len(problems)
    arr = (_FuseProblem * n)()
    keep, outs = [], []
    for i, pr in enumerate(problems):
        p = arr[i]
        F = dict(pr["train"])
        nt = len(F["x"])
        F.setdefault("angle", np.zeros(nt, np.float32)); F.setdefault("occupied", np.zeros(nt, np.uint8))
        _fill_train(p, F, keep)
        q = pr["query"]
        m = len(q["valid"])
        a = dict(q_valid=_arr(q["valid"], np.uint8), q_pos=_arr(q["pos"], np.float32).reshape(-1, 3), q_normal=_arr(q["normal"], np.float32).reshape(-1, 3),
                 q_min_dist=_arr(q["min_dist"], np.float32), q_max_dist=_arr(q["max_dist"], np.float32), q_desc=_arr(q["desc"], np.uint8).reshape(-1, 32))
        for k, v in a.items():
            setattr(p, k, v.ctypes.data)
        p.nq = m
        p.rcw = (ctypes.c_float * 9)(*np.asarray(pr["R"], np.float32).reshape(9))
        p.tcw = (ctypes.c_float * 3)(*np.asarray(pr["t"], np.float32).reshape(3))
        p.ow = (ctypes.c_float * 3)(*np.asarray(pr["ow"], np.float32).reshape(3))
        p.fx, p.fy, p.cx, p.cy, p.bf = [float(v) for v in pr["K5"]]
        p.bounds = (ctypes.c_double * 4)(*[float(v) for v in pr["bounds"]])
        p.scale_factors = (ctypes.c_float * 8)(*np.asarray(pr["scale_factors"], np.float32))
        p.inv_level_sigma2 = (ctypes.c_float * 8)(*np.asarray(pr["inv_level_sigma2"], np.float32))
        p.log_scale_factor = float(pr["log_scale_factor"]); p.n_levels = int(pr["n_levels"]); p.th = float(pr["th"])
        bi = np.full(max(m, 1), -1, np.int32); bd = np.full(max(m, 1), 256, np.int32)
        p.best_idx = bi.ctypes.data; p.best_dist = bd.ctypes.data
        keep.append(a); outs.append((bi, bd, m))
    check(lib.ps_fuse_search(self._h, arr, n))
    return [(bi[:m].copy(), bd[:m].copy()) for bi, bd, m in outs]


ORBmatcher.FuseSearch = _fuse_search
