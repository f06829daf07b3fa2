// Micro-benchmark (developer tool): v_mfma_f64_16x16x4_f64 on gfx950 as ba_solve_rs uses it - shader-clock cycles (s_memtime) per
// matrix instruction for (a) independent accumulators back to back, (b) one dependent chain, (c) chains of 4 on rotating
// accumulators (a rank-16 tile update), with 1, 2 and 4 waves per SIMD (blockDim 256 / 512 / 1024 on one CU), and (d) a dependent
// FP64 FMA chain on wave 0 while the other waves of the workgroup run (c): does the vector FP64 pipe wait behind the matrix cores?
//   hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64 && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* cyc, int iters, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  d4 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = d4{1.0 * lane, 2.0, 3.0, 4.0 + i};
  double a = 1.0 + lane * 1e-3, b = 0.999;
  __syncthreads();
  long long t0 = clock64();
  if (mode == 0) {          // 8 independent accumulators
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  } else if (mode == 1) {   // one dependent chain
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
  } else if (mode == 2) {   // chains of 4, then the next accumulator
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i / 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i / 4], 0, 0, 0);
  } else if (mode == 3) {   // two chains of 4 interleaved
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
  } else if (mode == 6) {   // four accumulators round-robin
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
  } else if (mode == 7) {   // three accumulators round-robin
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 9; i++) acc[i % 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % 3], 0, 0, 0);
  } else {                  // wave 0: dependent FMA chain; the others: mode 2
    if (wave == 0) {
      double c = a;
      for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) c = __fma_rn(c, b, a);
      acc[0][0] = c;
    } else if (mode == 4) {
      for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i / 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i / 4], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 8);
  const int iters = 2000;
  const char* names[] = {"8 independent accumulators", "one dependent chain", "chains of 4", "two chains of 4 interleaved",
                         "wave 0: dependent v_fma_f64 chain beside matrix waves", "wave 0: dependent v_fma_f64 chain alone",
                         "four accumulators round-robin", "three accumulators round-robin (9 per iteration)"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 8; mode++)
    for (int threads = 256; threads <= 1024; threads *= 2) {
      hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, iters, mode);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, iters, mode);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      long long h[16];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const int waves = threads / 64;
      printf("%-56s %d waves/SIMD: wave 0 %.1f cycles per instruction", names[mode], waves / 4, (double)h[0] / (iters * 8));
      if (mode < 4) printf(", per SIMD %.1f", (double)h[0] / (iters * 8) / (waves / 4));
      else printf(" (wave 1: %.1f)", (double)h[1] / (iters * 8));
      printf("   [kernel %.3f ms -> %.2f GHz shader clock if the loop is the kernel]\n", ms, (double)h[0] / (ms * 1e6));
    }
  return 0;
}
