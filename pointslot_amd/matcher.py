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

    def DescriptorDistanceMatrix(self, q, t):
        """bulk ORBmatcher::DescriptorDistance: uint16 [nq, nt]"""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), len(t)), np.uint16)
        check(lib.ps_hamming_matrix(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data))
        return out

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
