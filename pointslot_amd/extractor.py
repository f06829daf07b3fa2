"""Host-side mirror of ORB_SLAM2::ORBextractor (/root/reference/include/ORBextractor.h:51-85) on top
of the C-ABI.  Same constructor arguments, getters and call semantics; keypoints come back as a numpy
structured array that is byte-compatible with cv::KeyPoint, descriptors as an N x 32 uint8 matrix.
"""
import ctypes

import numpy as np

from ._lib import lib, check

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28
EDGE_THRESHOLD = 19


class _Config(ctypes.Structure):
    _fields_ = [("nfeatures", ctypes.c_int32), ("scale_factor", ctypes.c_float), ("nlevels", ctypes.c_int32),
                ("ini_th_fast", ctypes.c_int32), ("min_th_fast", ctypes.c_int32), ("max_batch", ctypes.c_int32),
                ("device", ctypes.c_int32)]


lib.ps_orb_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(ctypes.c_void_p)]
lib.ps_orb_destroy.argtypes = [ctypes.c_void_p]
lib.ps_orb_destroy.restype = None
lib.ps_orb_get_tables.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
lib.ps_orb_level_size.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
lib.ps_orb_extract.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                               ctypes.c_void_p]
lib.ps_orb_extract_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.ps_orb_extract_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
lib.ps_orb_batch_device_outputs.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
lib.ps_orb_batch_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int)]
lib.ps_orb_sync.argtypes = [ctypes.c_void_p]
lib.ps_orb_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
lib.ps_orb_stereo_match_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float]
lib.ps_orb_stereo_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
lib.ps_orb_stereo_match_keys.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.ps_orb_stereo_match_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.ps_orb_extract_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]


class _StereoFrame(ctypes.Structure):
    _fields_ = [("kps", ctypes.c_void_p), ("desc", ctypes.c_void_p), ("u_right", ctypes.c_void_p), ("depth", ctypes.c_void_p),
                ("cap", ctypes.c_int32), ("n", ctypes.c_int32), ("n_right", ctypes.c_int32), ("kept", ctypes.c_int32)]


lib.ps_orb_stereo_fetch_frames.argtypes = [ctypes.c_void_p, ctypes.POINTER(_StereoFrame), ctypes.c_int]
lib.ps_orb_enable_stage_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ps_orb_stage_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int)]


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) — ORBextractor.cc:410."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_batch=1, device=0,
                 keep_pyramid=False):
        self._h = ctypes.c_void_p()
        cfg = _Config(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_batch, device)
        check(lib.ps_orb_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.nfeatures, self.nlevels, self.max_batch = nfeatures, nlevels, max_batch
        self._scale_factor = float(np.float32(scaleFactor))
        self.keep_pyramid = keep_pyramid
        self.mvImagePyramid = []   # public member of the reference (ORBextractor.h:85)
        t = [np.zeros(nlevels, np.float32) for _ in range(4)] + [np.zeros(nlevels, np.int32)]
        check(lib.ps_orb_get_tables(self._h, *[a.ctypes.data for a in t]))
        self._tables = t
        self.capacity = nfeatures + 4 * nlevels + 64

    def close(self):
        if self._h:
            lib.ps_orb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters, ORBextractor.h:61-83
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self._scale_factor
    def GetScaleFactors(self): return self._tables[0].copy()
    def GetInverseScaleFactors(self): return self._tables[1].copy()
    def GetScaleSigmaSquares(self): return self._tables[2].copy()
    def GetInverseScaleSigmaSquares(self): return self._tables[3].copy()
    def features_per_level(self): return self._tables[4].copy()

    def level_size(self, w, h, level):
        wl, hl = ctypes.c_int32(), ctypes.c_int32()
        check(lib.ps_orb_level_size(self._h, w, h, level, ctypes.byref(wl), ctypes.byref(hl)))
        return wl.value, hl.value

    def detect_masked(self, image, mask):
        """The object features of a frame (Frame::ExtractObjORB, /root/reference/src/Frame.cc:2623-2665) through the declared
        stand-in for cv::ORB + mask (ps_orb_extract_masked): this extractor's pipeline, keypoints outside the mask (zero
        bytes) dropped before the quadtree.  Returns (keypoints, descriptors)."""
        image = np.ascontiguousarray(image); mask = np.ascontiguousarray(mask)
        if image.dtype != np.uint8 or image.ndim != 2 or mask.shape != image.shape or mask.dtype != np.uint8:
            raise AssertionError("image and mask must be CV_8UC1 of the same size")
        h, w = image.shape
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = ctypes.c_int(0)
        check(lib.ps_orb_extract_masked(self._h, image.ctypes.data, mask.ctypes.data, w, h, image.strides[0], mask.strides[0], kps.ctypes.data,
                                        desc.ctypes.data, self.capacity, ctypes.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def __call__(self, image, mask=None):
        """operator()(image, mask /*ignored*/, keypoints, descriptors).  Returns (keypoints, descriptors);
        descriptors is None when no keypoint was found (the reference releases the matrix)."""
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), None
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")   # ORBextractor.cc:1050
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = ctypes.c_int(0)
        planes, ptrs = None, None
        if self.keep_pyramid:
            planes = []
            for l in range(self.nlevels):
                wl, hl = self.level_size(w, h, l)
                planes.append(np.zeros((hl + 2 * EDGE_THRESHOLD, wl + 2 * EDGE_THRESHOLD), np.uint8))
            ptrs = (ctypes.c_void_p * self.nlevels)(*[p.ctypes.data for p in planes])
        check(lib.ps_orb_extract(self._h, image.ctypes.data, w, h, image.strides[0], kps.ctypes.data,
                                 desc.ctypes.data, self.capacity, ctypes.byref(n), ptrs))
        if planes is not None:
            e = EDGE_THRESHOLD
            self.mvImagePyramid = [p[e:-e, e:-e] for p in planes]
        if n.value == 0:
            return kps[:0], None
        return kps[:n.value].copy(), desc[:n.value].copy()

    # ---- batched device-resident path ----
    def extract_batch_device(self, d_ptr, nimg, w, h, stride, pitch, stream=None):
        check(lib.ps_orb_extract_batch_device(self._h, d_ptr, nimg, w, h, stride, pitch, stream))

    def extract_batch(self, images):
        """ps_orb_extract_batch: images of identical shape in host memory (a list of 2-D uint8 arrays); results stay on the
        device until fetch() / stereo_fetch_frames()."""
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        h, w = imgs[0].shape
        if any(im.shape != (h, w) for im in imgs):
            raise ValueError("all images of a batch must have the same size")
        ptrs = (ctypes.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        check(lib.ps_orb_extract_batch(self._h, ptrs, len(imgs), w, h, w))

    def sync(self):
        check(lib.ps_orb_sync(self._h))

    def fetch(self, image):
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = ctypes.c_int(0)
        check(lib.ps_orb_batch_fetch(self._h, image, kps.ctypes.data, desc.ctypes.data, self.capacity, ctypes.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def device_outputs(self):
        a, b, c, cap = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32()
        check(lib.ps_orb_batch_device_outputs(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(cap)))
        return a.value, b.value, c.value, cap.value

    def debug_read(self, image, level, what, w, h):
        wl, hl = self.level_size(w, h, level)
        n = ctypes.c_int(0)
        if what == 0:
            out = np.zeros((hl + 38, wl + 38), np.uint8)
        elif what == 1:
            out = np.zeros((hl, wl), np.uint8)
        else:
            out = np.zeros((wl * hl // 4 + 16, 3), np.int32)
        check(lib.ps_orb_debug_read(self._h, image, level, what, out.ctypes.data, out.nbytes, ctypes.byref(n)))
        return out if what < 2 else out[:n.value].copy()

    def enable_stage_timing(self, on=True):
        check(lib.ps_orb_enable_stage_timing(self._h, 1 if on else 0))

    def stage_times(self):
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_float * 16)()
        n = ctypes.c_int(0)
        check(lib.ps_orb_stage_times(self._h, names, ms, 16, ctypes.byref(n)))
        return {names[i].decode(): ms[i] for i in range(n.value)}

    # ---- Frame::ComputeStereoMatches on the device-resident results ----
    def stereo_match_batch(self, npairs, mb, mbf):
        """left/right interleaved in the last batch (image 2k = left, 2k+1 = right)"""
        check(lib.ps_orb_stereo_match_batch(self._h, npairs, mb, mbf))

    def stereo_fetch(self, pair):
        ur = np.zeros(self.capacity, np.float32); dp = np.zeros(self.capacity, np.float32)
        n = ctypes.c_int(0); kept = ctypes.c_int(0)
        check(lib.ps_orb_stereo_fetch(self._h, pair, ur.ctypes.data, dp.ctypes.data, self.capacity, ctypes.byref(n), ctypes.byref(kept)))
        return ur[:n.value].copy(), dp[:n.value].copy(), kept.value

    def stereo_fetch_frames(self, npairs):
        """ps_orb_stereo_fetch_frames: (keypoints, descriptors, mvuRight, mvDepth, kept) of every left image, one transfer."""
        bufs, frames = [], (_StereoFrame * npairs)()
        for k in range(npairs):
            b = (np.zeros(self.capacity, KEYPOINT_DTYPE), np.zeros((self.capacity, 32), np.uint8), np.zeros(self.capacity, np.float32),
                 np.zeros(self.capacity, np.float32))
            bufs.append(b)
            frames[k] = _StereoFrame(b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data, b[3].ctypes.data, self.capacity, 0, 0, 0)
        check(lib.ps_orb_stereo_fetch_frames(self._h, frames, npairs))
        return [(b[0][:f.n].copy(), b[1][:f.n].copy(), b[2][:f.n].copy(), b[3][:f.n].copy(), f.kept) for b, f in zip(bufs, frames)]


def ComputeStereoMatches(left, right, mb, mbf):
    """Frame::ComputeStereoMatches for two ORBextractor objects that have each processed one image.
    Returns (mvuRight, mvDepth) indexed like the left keypoints."""
    ur = np.zeros(left.capacity, np.float32); dp = np.zeros(left.capacity, np.float32)
    n = ctypes.c_int(0)
    check(lib.ps_orb_stereo_match_pair(left._h, right._h, mb, mbf, ur.ctypes.data, dp.ctypes.data, left.capacity, ctypes.byref(n)))
    return ur[:n.value].copy(), dp[:n.value].copy()


def ComputeObjStereoMatches(left, right, kps_l, desc_l, kps_r, desc_r, mb, mbf):
    """Frame::ComputeObjStereoMatches (Frame.cc:2318-2503): object key sets against the two extractors' pyramids.
    Returns (mvuTempObjKeysRight, mvTempObjDepth, kept)."""
    kl = np.ascontiguousarray(kps_l, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kps_r, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(desc_l, np.uint8).reshape(-1, 32); dr = np.ascontiguousarray(desc_r, np.uint8).reshape(-1, 32)
    ur = np.full(max(len(kl), 1), -1.0, np.float32); dp = np.full(max(len(kl), 1), -1.0, np.float32)
    kept = ctypes.c_int(0)
    check(lib.ps_orb_stereo_match_keys(left._h, right._h, kl.ctypes.data, dl.ctypes.data, len(kl), kr.ctypes.data, dr.ctypes.data, len(kr),
                                       mb, mbf, ur.ctypes.data, dp.ctypes.data, ctypes.byref(kept)))
    return ur[:len(kl)], dp[:len(kl)], kept.value
