// Micro-benchmark (developer tool): what one wavefront alone on a SIMD pays per FP64 instruction on gfx950 - independent FMAs,
// a dependent FMA chain, FMAs whose scalar operand comes from v_readlane, and the rcp + Newton reciprocal.  Prints shader-clock
// cycles per instruction (s_memtime based).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast fp64_issue.hip -o fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__global__ void k(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  double a[24], x = 1.0 + lane * 1e-3, y = 0.999;
  for (int i = 0; i < 24; i++) a[i] = i + lane;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 24; i++) a[i] = __fma_rn(a[i], y, x);          // 24 independent FMAs
  long long t1 = clock64();
  double c = x;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 24; i++) c = __fma_rn(c, y, x);                // dependent chain
  long long t2 = clock64();
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 24; i++) a[i] = __fma_rn(-x, rl(a[(i + 1) % 24], i), a[i]);   // FMA with a broadcast operand
  long long t3 = clock64();
  for (int it = 0; it < iters; it++) {   // the same with every broadcast fetched before the first FMA
    double b[24];
#pragma unroll
    for (int i = 0; i < 24; i++) b[i] = rl(a[(i + 1) % 24], i);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 24; i++) a[i] = __fma_rn(-x, b[i], a[i]);
  }
  long long t3b = clock64();
  __shared__ __attribute__((aligned(16))) double lbuf[64];
  lbuf[lane & 63] = x;
  __syncthreads();
  long long t3c = clock64();
  typedef double d2 __attribute__((ext_vector_type(2)));
  for (int it = 0; it < iters; it++) {   // FMAs whose operand is an LDS broadcast, two values per 128-bit read
#pragma unroll
    for (int i = 0; i < 24; i += 2) {
      const d2 v = *reinterpret_cast<volatile d2*>(&lbuf[i]);
      a[i] = __fma_rn(-x, v.x, a[i]);
      a[i + 1] = __fma_rn(-x, v.y, a[i + 1]);
    }
  }
  long long t3d = clock64();
  double r = x;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) { double q = __builtin_amdgcn_rcp(r); q = __fma_rn(__fma_rn(-r, q, 1.0), q, q); q = __fma_rn(__fma_rn(-r, q, 1.0), q, q); r = q + 1.5; }
  long long t4 = clock64();
  double s = c + r;
  for (int i = 0; i < 24; i++) s += a[i];
  out[blockIdx.x * blockDim.x + lane] = s;
  if (lane == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3d; cyc[4] = t3b - t3; cyc[5] = t3d - t3c; }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 1024 * 8); hipMalloc(&cyc, 64);
  const int iters = 200;
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, iters);   // `waves` wavefronts per SIMD on one CU
    hipDeviceSynchronize();
    long long h[6]; hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost);
    printf("%d wave(s) per SIMD: independent FMA %.1f, dependent FMA %.1f, FMA + 2 readlane %.1f (per FMA), rcp + 2 Newton + add %.1f, broadcasts first then FMAs %.1f, FMA with LDS broadcast operand (b128 per two) %.1f clock64 ticks\n", waves,
           h[0] / (24.0 * iters), h[1] / (24.0 * iters), h[2] / (24.0 * iters), h[3] / (8.0 * iters), h[4] / (24.0 * iters), h[5] / (24.0 * iters));
  }
  return 0;
}
