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


def distribute(keys_xyz, minX, maxX, minY, maxY, N):
    keys = np.ascontiguousarray(keys_xyz, np.int32)
    out = np.zeros((N + 16 + 64, 3), np.int32)
    n = lib().orc_distribute(keys.ctypes.data, len(keys), minX, maxX, minY, maxY, N, out.ctypes.data)
    return out[:n]
