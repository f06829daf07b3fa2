"""Host-side mirror of the object-feature detector: OpenCV's own ORB as Frame::ExtractObjORB / OpencvORBDetector use it
(/root/reference/src/Frame.cc:2623-2627: cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, kp, descriptor)),
restated for the GPU behind ps_cvorb_* (unverifiable against OpenCV in the build image)."""
import ctypes

import numpy as np

from ._lib import lib, check
from .extractor import KEYPOINT_DTYPE

lib.ps_cvorb_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
lib.ps_cvorb_destroy.argtypes = [ctypes.c_void_p]
lib.ps_cvorb_destroy.restype = None
lib.ps_cvorb_detect_and_compute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.ps_cvorb_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
lib.ps_cvorb_detect_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
lib.ps_cvorb_batch_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]


class ORB:
    """cv::ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold) with OpenCV's other defaults (firstLevel 0, WTA_K 2,
    HARRIS_SCORE, patchSize 31, fastThreshold 20)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, edgeThreshold=19, fastThreshold=20, device=0):
        self._h = ctypes.c_void_p()
        check(lib.ps_cvorb_create(nfeatures, scaleFactor, nlevels, edgeThreshold, fastThreshold, device, ctypes.byref(self._h)))
        self.nlevels = nlevels
        self.capacity = 16 * nfeatures + 4096

    def close(self):
        if self._h:
            lib.ps_cvorb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detectAndCompute(self, image, mask=None):
        image = np.ascontiguousarray(image)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")
        h, w = image.shape
        mp, ms = None, 0
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            assert mask.shape == image.shape and mask.dtype == np.uint8
            mp, ms = mask.ctypes.data, mask.strides[0]
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = ctypes.c_int(0)
        check(lib.ps_cvorb_detect_and_compute(self._h, image.ctypes.data, mp, w, h, image.strides[0], ms, kps.ctypes.data, desc.ctypes.data,
                                              self.capacity, ctypes.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def detect_batch_device(self, d_imgs, d_masks, nimg, w, h, stride=None, image_pitch=None, mask_stride=None, mask_pitch=None, stream=None):
        """The batched, device-resident form: `nimg` images and masks in HBM (device pointers); asynchronous."""
        stride = stride or w; mask_stride = mask_stride or w
        check(lib.ps_cvorb_detect_batch_device(self._h, d_imgs, d_masks, nimg, w, h, stride, image_pitch or stride * h, mask_stride,
                                               mask_pitch or mask_stride * h, stream))

    def batch_fetch(self, image):
        kps = np.zeros(2048, KEYPOINT_DTYPE)
        desc = np.zeros((2048, 32), np.uint8)
        n = ctypes.c_int(0)
        check(lib.ps_cvorb_batch_fetch(self._h, image, kps.ctypes.data, desc.ctypes.data, 2048, ctypes.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, level):
        out = np.zeros(2, np.int32)
        check(lib.ps_cvorb_debug_read(self._h, level, 4, out.ctypes.data, 8, None))
        return int(out[0]), int(out[1])

    def debug_plane(self, level, what):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(lib.ps_cvorb_debug_read(self._h, level, what, out.ctypes.data, out.nbytes, None))
        return out

    def debug_fast(self, level):
        w, h = self.level_size(level)
        out = np.zeros((((w + 1) // 2) * ((h + 1) // 2), 4), np.float32)
        n = ctypes.c_int(0)
        check(lib.ps_cvorb_debug_read(self._h, level, 3, out.ctypes.data, out.nbytes, ctypes.byref(n)))
        return out[:n.value].copy()
