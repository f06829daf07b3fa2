// Shared host-side helpers of libpointslot_hip.so: error reporting for the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pointslot_hip.h"

// Records a formatted message (thread-local) and returns `code`.
int ps_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define PS_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return ps_set_error(PS_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)
