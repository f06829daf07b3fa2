"""Runs the known-byte streaming kernels of the library (ps_debug_traffic_kernel) - to be wrapped in rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE by tools/pmc_traffic.sh, which turns the counter values into calibration factors."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd._lib import lib, check  # noqa: E402

lib.ps_debug_traffic_kernel.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
BYTES = 1 << 30      # 1 GiB: four times the Infinity Cache
for mode in range(4):
    check(lib.ps_debug_traffic_kernel(0, mode, BYTES, 3))
print("calibration kernels done: %d bytes per launch" % BYTES)
