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
// FP64 matrix-core rate: every wave keeps 8 independent accumulator tiles in flight (v_mfma_f64_16x16x4_f64: 2 * 16 * 16 * 4 flop per
// instruction and wave), no memory traffic inside the loop
typedef double ps_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_f64_loop(double* sink, int iters) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  ps_d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = ps_d4{(double)i, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678) sink[0] = s;   // keeps the loop alive
}
// Known-byte streaming kernels for calibrating the PMC byte counters (MI355X_MICROARCH.md, section HBM: FETCH_SIZE / WRITE_SIZE
// are only calibrated for 16 B / lane reads; "calibrate on a known byte count in your own access pattern"): every lane reads (or
// writes) `words` consecutive 4-byte or 16-byte elements of a buffer exactly once, wave-coalesced.
template <typename T> __global__ __launch_bounds__(256) void traffic_read(const T* src, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = src[i];
    acc ^= ((const uint32_t*)&v)[0];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <typename T> __global__ __launch_bounds__(256) void traffic_write(T* dst, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v;
    for (size_t k = 0; k < sizeof(T) / 4; k++) ((uint32_t*)&v)[k] = seed + (uint32_t)i;
    dst[i] = v;
  }
}
}  // namespace

extern "C" {
int ps_debug_traffic_kernel(int device, int mode, size_t bytes, int repeats) {
  PS_HIP(hipSetDevice(device));
  if (mode < 0 || mode > 3 || bytes < 1024 || repeats < 1) return ps_set_error(PS_ERR_INVALID, "ps_debug_traffic_kernel: mode 0..3, bytes >= 1024");
  bytes &= ~(size_t)1023;
  uint8_t* buf = nullptr;
  uint32_t* sink = nullptr;
  PS_HIP(hipMalloc(&buf, bytes));
  PS_HIP(hipMalloc(&sink, 4));
  PS_HIP(hipMemset(buf, 1, bytes));
  PS_HIP(hipDeviceSynchronize());
  const int blocks = 256 * 16;
  for (int r = 0; r < repeats; r++) {
    if (mode == 0) hipLaunchKernelGGL(traffic_read<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    else if (mode == 1) hipLaunchKernelGGL(traffic_read<uint32_t>, dim3(blocks), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, sink);
    else if (mode == 2) hipLaunchKernelGGL(traffic_write<uint32_t>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4, (uint32_t)r);
    else hipLaunchKernelGGL(traffic_write<uint4>, dim3(blocks), dim3(256), 0, 0, (uint4*)buf, bytes / 16, (uint32_t)r);
  }
  PS_HIP(hipGetLastError());
  PS_HIP(hipDeviceSynchronize());
  hipFree(buf); hipFree(sink);
  return PS_OK;
}
int ps_debug_mfma_f64_peak(int device, double* tflops) {
  if (!tflops) return ps_set_error(PS_ERR_INVALID, "null argument");
  PS_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PS_HIP(hipGetDeviceProperties(&prop, device));
  double* sink = nullptr;
  PS_HIP(hipMalloc(&sink, 8));
  hipEvent_t e0, e1;
  PS_HIP(hipEventCreate(&e0)); PS_HIP(hipEventCreate(&e1));
  const int blocks = prop.multiProcessorCount * 8, iters = 4096;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(mfma_f64_loop, dim3(blocks), dim3(256), 0, 0, sink, 64);   // warm-up (code load, clocks)
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    PS_HIP(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_f64_loop, dim3(blocks), dim3(256), 0, 0, sink, iters);
    PS_HIP(hipEventRecord(e1, 0));
    PS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PS_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)blocks * 4 /*waves*/ * iters * 8 * 2048.0;
    if (ms > 0 && flop / (ms * 1e-3) / 1e12 > best) best = flop / (ms * 1e-3) / 1e12;
  }
  PS_HIP(hipGetLastError());
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(sink);
  *tflops = best;
  return PS_OK;
}
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
