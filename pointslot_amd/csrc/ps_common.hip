// Error reporting and device queries of the C-ABI (include/pointslot_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "ps_common.h"

static thread_local char g_err[512] = "";

int ps_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {
const char* ps_last_error(void) { return g_err; }
const char* ps_version(void) { return "pointslot_hip 0.1 (gfx950)"; }
int ps_device_count(int* count) {
  if (!count) return ps_set_error(PS_ERR_INVALID, "null argument");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return ps_set_error(PS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return PS_OK;
}
}
