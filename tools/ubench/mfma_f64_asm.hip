// Micro-benchmark (developer tool): the inline-asm rank-16 tile update of ba_solve_rs (4 dependent v_mfma_f64_16x16x4_f64 on one
// accumulator tile) in isolation: cycles per block of 4 with the accumulator in a[] or v[] registers, with and without the
// trailing s_nop, 12 tiles round-robin, one wave per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f64_asm.hip -o mfma_f64_asm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define BLOCK(CONSTRAINT, C, NOPS) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %3, %4, %0\n\t" \
  "v_mfma_f64_16x16x4_f64 %0, %5, %6, %0\n\tv_mfma_f64_16x16x4_f64 %0, %7, %8, %0" NOPS : CONSTRAINT(C) : "v"(a), "v"(b), "v"(a2), "v"(b2), "v"(a), "v"(b2), "v"(a2), "v"(b))
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  d4 acc[12];
  for (int i = 0; i < 12; i++) acc[i] = d4{1.0 * lane, 2.0, 3.0, 4.0 + i};
  double a = 1.0 + lane * 1e-3, b = 0.999, a2 = 0.5 + lane * 1e-4, b2 = 1.001;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
      if (mode == 0) BLOCK("+a", acc[i], "\n\ts_nop 15\n\ts_nop 3");
      else if (mode == 1) BLOCK("+a", acc[i], "");
      else if (mode == 2) BLOCK("+v", acc[i], "\n\ts_nop 15\n\ts_nop 3");
      else if (mode == 3) BLOCK("+v", acc[i], "");
      else if (mode == 4) {   // the same through the builtin
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0); acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[i], 0, 0, 0); acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc[i], 0, 0, 0);
      } else if (mode == 5) BLOCK("+a", acc[0], "");   // one tile over and over
      else {   // builtin, the K-steps of two tiles interleaved
        if (i & 1) continue;
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0); acc[i + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i + 1], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[i], 0, 0, 0); acc[i + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[i + 1], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[i], 0, 0, 0); acc[i + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[i + 1], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc[i], 0, 0, 0); acc[i + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc[i + 1], 0, 0, 0);
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 12; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 16 * 8);
  const int iters = 500;
  const char* names[] = {"a[] accumulators, s_nop 15 + 3 after the block", "a[] accumulators, no nops", "v[] accumulators, nops", "v[] accumulators, no nops", "builtin", "asm, one tile over and over", "builtin, two tiles interleaved"};
  for (int mode = 0; mode < 7; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, cyc, iters, mode);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, cyc, iters, mode);
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-52s %.1f cycles per block of 4 (wave 0), %.1f (wave 3)\n", names[mode], (double)h[0] / (iters * 12), (double)h[3] / (iters * 12));
  }
  return 0;
}
