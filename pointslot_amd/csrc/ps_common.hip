// Error reporting and device queries of the C-ABI (include/pointslot_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include "ps_common.h"

static thread_local char g_err[512] = "";

int ps_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {
// fills the whole LDS allocation of every resident workgroup with a byte pattern and leaves it there: the next kernels on those
// CUs find it in whatever LDS they do not initialise themselves
__global__ void poison_lds(uint32_t pattern, int words, uint32_t* sink) {
  extern __shared__ uint32_t lds_words[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds_words[i] = pattern;
  __syncthreads();
  if (sink && lds_words[(threadIdx.x * 97) % words] != pattern) sink[0] = 1;   // keeps the stores alive
}
}  // namespace

extern "C" {
int ps_debug_poison_lds(int device, uint32_t pattern) {
  PS_HIP(hipSetDevice(device));
  const int bytes = 64 * 1024, words = bytes / 4;
  // two workgroups of 64 KB per CU and several rounds, so that every part of every CU's LDS is visited
  hipLaunchKernelGGL(poison_lds, dim3(256 * 8), dim3(256), bytes, 0, pattern, words, (uint32_t*)nullptr);
  PS_HIP(hipGetLastError());
  PS_HIP(hipDeviceSynchronize());
  return PS_OK;
}
void* ps_pinned_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    ps_set_error(PS_ERR_HIP, "ps_pinned_alloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}
void ps_pinned_free(void* p) {
  if (p) (void)hipHostFree(p);
}
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
