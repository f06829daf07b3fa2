// Micro-benchmark (developer tool): THROUGHPUT of the vector instructions the kernels are made of, on gfx950: W waves per SIMD on every CU run 16
// independent chains of one instruction; wall time by HIP events -> cycles per wave-instruction and SIMD at the clock given as 2nd argument
// (a full-rate wave64 instruction takes 4).   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate;  ./valu_rate [W=8] [GHz=2.4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N 16
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int OP> __device__ __forceinline__ uint32_t op(uint32_t x, uint32_t y) {
  if (OP == 0) return x + y;
  if (OP == 1) return x * y;
  if (OP == 2) return __umulhi(x, y);
  if (OP == 3) return (uint32_t)__mul24((int)x, (int)y);
  if (OP == 4) return (uint32_t)(__mul24((int)x, (int)y) + (int)x);
  if (OP == 5) return __builtin_amdgcn_perm(x, y, 0x07050301u);
  if (OP == 6) return __builtin_amdgcn_alignbyte(x, y, 1);
  if (OP == 7) return __builtin_amdgcn_udot4(x, y, x, false);
  if (OP == 8) return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y), x, false);
  if (OP == 9) return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y ^ x)));
  if (OP == 10) return __builtin_amdgcn_sad_u16(x, y, x);
  if (OP == 11) return (x << 3) + y;
  if (OP == 12) return (uint32_t)__popc(x) + y;
  if (OP == 13) return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  if (OP == 14) { uint32_t r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
  if (OP == 15) { uint32_t r; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(x + y)); return r; }
  if (OP == 16) { typedef float f2 __attribute__((ext_vector_type(2))); const f2 f = __builtin_amdgcn_cvt_pk_f32_fp8((int)(x & 0x57575757u), false); return __float_as_uint(f.x) ^ __float_as_uint(f.y) ^ y; }
  if (OP == 17) return __float_as_uint((float)(int8_t)(x & 0xFF)) ^ y;
  return x;
}
template <int OP> __global__ void k(uint32_t* out, int iters, uint32_t seed, uint32_t y) {
  uint32_t a[N];
  for (int i = 0; i < N; i++) a[i] = seed + threadIdx.x + i * 7919u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = op<OP>(a[i], y);
  }
  uint32_t s = 0;
  for (int i = 0; i < N; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void kf64(uint32_t* out, int iters, double y) {
  double a[N];
  for (int i = 0; i < N; i++) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = __builtin_fma(a[i], y, 1e-3);
  }
  double s = 0;
  for (int i = 0; i < N; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double2loint(s);
}
template <int OP> float run(uint32_t* out, int grid, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, 77u);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, 77u);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 8; const double ghz = argc > 2 ? atof(argv[2]) : 2.4;
  const int grid = 256 * W, iters = 4000;
  uint32_t* out; (void)hipMalloc(&out, (size_t)grid * 256 * 4);
  const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_perm_b32", "v_alignbyte_b32", "v_dot4_u32_u8", "v_dot2_u32_u16",
                         "v_pk_max_u16 (+xor)", "v_sad_u16", "v_lshl_add_u32", "v_bcnt + add", "v_add_u32 dpp", "v_mul_hi_u32_u24", "v_sat_pk_u8_i16 (+add)", "v_cvt_pk_f32_fp8 (+and, 2 xor)", "v_cvt_f32_i32 sdwa (+and?, xor)"};
  float ms[19];
  ms[0] = run<0>(out, grid, iters); ms[1] = run<1>(out, grid, iters); ms[2] = run<2>(out, grid, iters); ms[3] = run<3>(out, grid, iters);
  ms[4] = run<4>(out, grid, iters); ms[5] = run<5>(out, grid, iters); ms[6] = run<6>(out, grid, iters); ms[7] = run<7>(out, grid, iters);
  ms[8] = run<8>(out, grid, iters); ms[9] = run<9>(out, grid, iters); ms[10] = run<10>(out, grid, iters); ms[11] = run<11>(out, grid, iters);
  ms[12] = run<12>(out, grid, iters); ms[13] = run<13>(out, grid, iters); ms[14] = run<14>(out, grid, iters); ms[15] = run<15>(out, grid, iters); ms[17] = run<16>(out, grid, iters); ms[18] = run<17>(out, grid, iters);
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kf64, dim3(grid), dim3(256), 0, 0, out, iters, 0.999);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kf64, dim3(grid), dim3(256), 0, 0, out, iters, 0.999);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms[16], e0, e1);
  }
  printf("%d waves per SIMD, %d x 16 instructions per wave, %.2f GHz assumed\n", W, iters, ghz);
  for (int i = 0; i < 19; i++)
    printf("%-32s %7.3f ms  %5.2f cycles per chain step and SIMD\n", i < 16 ? names[i] : i == 16 ? "v_fma_f64" : names[i - 1], ms[i], ms[i] * 1e-3 * ghz * 1e9 / ((double)iters * N * W));
  return 0;
}
