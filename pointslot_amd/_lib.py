"""ctypes binding of libpointslot_hip.so (the C-ABI declared in include/pointslot_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing the import
fails loudly; if no MI355X is visible every create call fails with PS_ERR_NO_DEVICE.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PS_LIB_PATH") or os.path.join(_HERE, "libpointslot_hip.so")   # PS_LIB_PATH: developer builds of the same library

PS_OK = 0
PS_ERR_INVALID = -1
PS_ERR_HIP = -2
PS_ERR_CAPACITY = -3
PS_ERR_NO_DEVICE = -4


class PointslotError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pointslot_hip error %d: %s" % (code, msg))
        self.code = code


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.  Two HIP runtimes in one
    process cannot both own the GPU, so when torch is installed bind to ITS runtime (same soname): load
    it first and the NEEDED entry of libpointslot_hip.so resolves to the already-loaded object; a later
    `import torch` then shares it too.  Without torch the system /opt/rocm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C pointslot_amd/csrc` (or __graft_entry__.build()); "
            "pointslot_amd has no CPU fallback" % LIB_PATH)
    _preload_torch_hip_runtime()
    return ctypes.CDLL(LIB_PATH)


lib = _load()
lib.ps_last_error.restype = ctypes.c_char_p
lib.ps_version.restype = ctypes.c_char_p


def check(rc):
    if rc != PS_OK:
        raise PointslotError(rc, lib.ps_last_error().decode("utf-8", "replace"))
    return rc


def device_count():
    n = ctypes.c_int(0)
    rc = lib.ps_device_count(ctypes.byref(n))
    return n.value if rc == PS_OK else 0


def poison_lds(pattern=0xFFFFFFFF, device=0):
    """Test / diagnostic: leave `pattern` in the LDS of every CU (see ps_debug_poison_lds)."""
    lib.ps_debug_poison_lds.argtypes = [ctypes.c_int, ctypes.c_uint32]
    check(lib.ps_debug_poison_lds(device, pattern))


lib.ps_pinned_alloc.restype = ctypes.c_void_p
lib.ps_pinned_alloc.argtypes = [ctypes.c_size_t]
lib.ps_pinned_free.restype = None
lib.ps_pinned_free.argtypes = [ctypes.c_void_p]


class PinnedBuffer:
    """Page-locked host memory (ps_pinned_alloc) as a numpy uint8 array: image buffers uploaded from it do not stall the
    caller (ps_orb_extract_batch).  Free with close() or let it go out of scope."""

    def __init__(self, nbytes):
        import numpy as np
        self._p = lib.ps_pinned_alloc(nbytes)
        if not self._p:
            raise PointslotError(-2, lib.ps_last_error().decode("utf-8", "replace"))
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(self._p))

    def close(self):
        if self._p:
            self.array = None
            lib.ps_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
