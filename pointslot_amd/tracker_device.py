"""The device-resident lockstep tracker (include/pointslot_hip.h: ps_tracker_*): many independent stereo sequences advance
one frame per call and the whole per-frame chain of the tracking thread — Frame::Frame, TrackWithMotionModel, TrackLocalMap
(/root/reference/src/Tracking.cc:2840-3160), the slice `tracker.StereoOdometry` drives call by call — is queued on one
stream with nothing returning to the host.  This class is a ctypes mirror of the C++ host class StereoOdometryDevice
(pointslot_amd/host/StereoOdometry.h)."""
import ctypes

import numpy as np

from ._lib import lib, check


class _Config(ctypes.Structure):
    _fields_ = [("n_sequences", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float), ("bf", ctypes.c_float),
                ("th_depth", ctypes.c_float), ("nfeatures", ctypes.c_int32), ("scale_factor", ctypes.c_float), ("nlevels", ctypes.c_int32),
                ("ini_th_fast", ctypes.c_int32), ("min_th_fast", ctypes.c_int32), ("max_steps", ctypes.c_int32), ("device", ctypes.c_int32),
                ("max_objects", ctypes.c_int32), ("max_map_objects", ctypes.c_int32)]


STAT_DTYPE = np.dtype([("state", "<i4"), ("tracked", "<i4"), ("n", "<i4"), ("mm_matches", "<i4"), ("retried", "<i4"), ("matches", "<i4"),
                       ("map_matches", "<i4"), ("lm_candidates", "<i4"), ("lm_inliers", "<i4"), ("overflowed", "<i4"), ("reserved", "<i4", 2)])
assert STAT_DTYPE.itemsize == 48
# ps_detection / ps_object_stat (include/pointslot_hip.h)
DETECTION_DTYPE = np.dtype([("id", "<i4"), ("bbox", "<i4", 4), ("reserved", "<i4", 3), ("scale", "<f8", 3), ("pose7", "<f8", 7)])
OBJECT_STAT_DTYPE = np.dtype([("id", "<i4"), ("n", "<i4"), ("stereo", "<i4"), ("tracked", "<i4"), ("is_new", "<i4"), ("track_ok", "<i4"), ("inliers", "<i4"),
                              ("bf_matches", "<i4"), ("lm_candidates", "<i4"), ("lm_matches", "<i4"), ("map_points", "<i4"), ("reinit", "<i4"),
                              ("dynamic", "<i4"), ("mo_dynamic", "<i4"), ("dyn_n_mono", "<i4"), ("dyn_n_stereo", "<i4"), ("tco", "<f8", 7),
                              ("dyn_mono", "<f8"), ("dyn_stereo", "<f8")])
assert DETECTION_DTYPE.itemsize == 112 and OBJECT_STAT_DTYPE.itemsize == 136


def pack_detections(dets_per_sequence, max_objects):
    """[S][max_objects] ps_detection from lists of object_tracker.detection_from_label dicts (unused slots: id = -1)."""
    out = np.zeros((len(dets_per_sequence), max_objects), DETECTION_DTYPE)
    out["id"] = -1
    for s, dets in enumerate(dets_per_sequence):
        assert len(dets) <= max_objects, "more detections in a frame than the tracker was created for"
        for j, d in enumerate(dets):
            out[s, j]["id"] = d["id"]; out[s, j]["bbox"] = d["bbox"]; out[s, j]["scale"] = d["scale"]; out[s, j]["pose7"] = d["pose7"]
    return out


lib.ps_tracker_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(ctypes.c_void_p)]
lib.ps_tracker_destroy.argtypes = [ctypes.c_void_p]
lib.ps_tracker_destroy.restype = None
lib.ps_tracker_step_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
lib.ps_tracker_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.ps_tracker_step_slot_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
lib.ps_tracker_fetch_objects.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.ps_tracker_sync.argtypes = [ctypes.c_void_p]
lib.ps_tracker_steps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
lib.ps_tracker_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.ps_tracker_reset.argtypes = [ctypes.c_void_p]
lib.ps_tracker_debug_set_overflow.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
lib.ps_tracker_enable_stage_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ps_tracker_stage_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.ps_tracker_orb.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]


class LockstepTracker:
    def __init__(self, n_sequences, K, bf, width, height, max_steps, th_depth=35.0, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=5,
                 device=0, max_objects=0, max_map_objects=0):
        fx, fy, cx, cy = [float(v) for v in K]
        cfg = _Config(n_sequences, width, height, fx, fy, cx, cy, float(bf), float(th_depth), nfeatures, scale, nlevels, ini_th, min_th,
                      max_steps, device, max_objects, max_map_objects)
        self._h = ctypes.c_void_p()
        check(lib.ps_tracker_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.n_sequences, self.width, self.height, self.max_objects = n_sequences, width, height, max_objects

    def close(self):
        if self._h:
            lib.ps_tracker_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step_device(self, d_ptr, stride=None, pitch=None):
        """One stereo frame of every sequence from images in HBM: sequence k's left image at d_ptr + 2k * pitch, right one
        pitch further.  Returns when the step is queued."""
        stride = self.width if stride is None else stride
        pitch = stride * self.height if pitch is None else pitch
        check(lib.ps_tracker_step_device(self._h, ctypes.c_void_p(d_ptr), stride, pitch))

    def step_slot_device(self, d_imgs, d_masks, d_dets, stride=None, pitch=None, mask_stride=None, mask_pitch=None):
        """One SLOT.MODE 4 frame of every sequence, camera chain + object chain: images as for step_device, the left 8-bit id masks
        (sequence k at d_masks + k * mask_pitch) and [S][max_objects] ps_detection records, all device pointers."""
        stride = self.width if stride is None else stride
        pitch = stride * self.height if pitch is None else pitch
        mask_stride = self.width if mask_stride is None else mask_stride
        mask_pitch = mask_stride * self.height if mask_pitch is None else mask_pitch
        check(lib.ps_tracker_step_slot_device(self._h, ctypes.c_void_p(d_imgs), stride, pitch, ctypes.c_void_p(d_masks), mask_stride, mask_pitch,
                                              ctypes.c_void_p(d_dets)))

    def fetch_objects(self, first=0, n=None):
        """[n, S, max_objects] OBJECT_STAT_DTYPE"""
        n = self.steps() - first if n is None else n
        out = np.zeros((n, self.n_sequences, self.max_objects), OBJECT_STAT_DTYPE)
        check(lib.ps_tracker_fetch_objects(self._h, first, n, out.ctypes.data))
        return out

    def step(self, left, right):
        """left / right: lists of n_sequences contiguous uint8 [h, w] arrays (host memory; pinned buffers upload asynchronously)."""
        assert len(left) == self.n_sequences and len(right) == self.n_sequences
        for a in list(left) + list(right):
            if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.shape == (self.height, self.width) and a.flags["C_CONTIGUOUS"]):
                raise ValueError("step(): every image must be a C-contiguous uint8 array of shape (%d, %d)" % (self.height, self.width))
        # pinned buffers are uploaded asynchronously: the arrays must stay alive and unchanged until the step has run
        self._in_flight = (list(left), list(right))
        pl = (ctypes.c_void_p * self.n_sequences)(*[a.ctypes.data for a in left])
        pr = (ctypes.c_void_p * self.n_sequences)(*[a.ctypes.data for a in right])
        check(lib.ps_tracker_step(self._h, pl, pr, self.width))

    def sync(self):
        check(lib.ps_tracker_sync(self._h))

    def steps(self):
        n = ctypes.c_int(0)
        check(lib.ps_tracker_steps(self._h, ctypes.byref(n)))
        return n.value

    def fetch(self, first=0, n=None):
        """(tcw [n, S, 4, 4] float32 — zeros where a frame has no pose, stats [n, S] STAT_DTYPE)"""
        n = self.steps() - first if n is None else n
        tcw = np.zeros((n, self.n_sequences, 4, 4), np.float32)
        st = np.zeros((n, self.n_sequences), STAT_DTYPE)
        check(lib.ps_tracker_fetch(self._h, first, n, tcw.ctypes.data, st.ctypes.data))
        return tcw, st

    def reset(self):
        check(lib.ps_tracker_reset(self._h))

    def enable_stage_timing(self, on=True):
        check(lib.ps_tracker_enable_stage_timing(self._h, 1 if on else 0))
        orb = ctypes.c_void_p()
        check(lib.ps_tracker_orb(self._h, ctypes.byref(orb)))
        check(lib.ps_orb_enable_stage_timing(orb, 1 if on else 0))

    def stage_times(self):
        """{stage: ms per step} of the chain, and the extractor's per-kernel stage times underneath ("orb/<kernel>")."""
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_float * 16)()
        n = ctypes.c_int(0)
        check(lib.ps_tracker_stage_times(self._h, names, ms, 16, ctypes.byref(n)))
        out = {names[i].decode(): float(ms[i]) for i in range(n.value)}
        orb = ctypes.c_void_p()
        check(lib.ps_tracker_orb(self._h, ctypes.byref(orb)))
        from . import extractor  # noqa: F401  (argtypes of ps_orb_stage_times)
        check(lib.ps_orb_stage_times(orb, names, ms, 16, ctypes.byref(n)))
        for i in range(n.value):
            out["orb/" + names[i].decode()] = float(ms[i])
        return out
