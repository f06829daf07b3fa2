// Micro-benchmark (developer tool): issue cost of the integer / packed instructions the ORB kernels are made of, on gfx950.
// One wave alone on a SIMD runs 16 independent chains of each instruction; prints shader-clock cycles per instruction
// (a full-rate wave64 VALU instruction takes 4).   hipcc --offload-arch=gfx950 -O3 int_issue.hip -o int_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 16
__device__ __forceinline__ uint32_t mulhi24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define BENCH(name, expr)                                                    \
  {                                                                          \
    uint32_t a[N];                                                           \
    for (int i = 0; i < N; i++) a[i] = seed + i * 7919u;                     \
    long long t0 = clock64();                                                \
    for (int it = 0; it < iters; it++) {                                     \
      _Pragma("unroll") for (int i = 0; i < N; i++) { uint32_t x = a[i]; a[i] = (expr); } \
    }                                                                        \
    long long t1 = clock64();                                                \
    uint32_t s = 0;                                                          \
    for (int i = 0; i < N; i++) s ^= a[i];                                   \
    sink ^= s;                                                               \
    if (lane == 0 && blockIdx.x == 0) cyc[nb] = t1 - t0;                     \
    nb++;                                                                    \
  }
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t* out, long long* cyc, int iters, uint32_t seed, uint32_t y) {
  const int lane = threadIdx.x;
  uint32_t sink = 0;
  int nb = 0;
  seed += lane;
  BENCH("v_add_u32", x + y)
  BENCH("v_mul_lo_u32", x * y)
  BENCH("v_mul_hi_u32", __umulhi(x, y))
  BENCH("v_mul_u32_u24", (uint32_t)__mul24((int)x, (int)y))
  BENCH("v_mul_hi_u32_u24", mulhi24(x, y))
  BENCH("v_mad_u32_u24", (uint32_t)(__mul24((int)x, (int)y) + (int)x))
  BENCH("v_perm_b32", __builtin_amdgcn_perm(x, y, 0x07050301u))
  BENCH("v_alignbyte", __builtin_amdgcn_alignbyte(x, y, 1))
  BENCH("v_dot4_u32_u8", __builtin_amdgcn_udot4(x, y, x, false))
  BENCH("v_dot2_u32_u16", __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y), x, false))
  BENCH("v_sad_u16", __builtin_amdgcn_sad_u16(x, y, x))
  BENCH("v_pk_max_u16", __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y))))
  BENCH("v_pk_sub_i16", __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, x) - __builtin_bit_cast(us2, y)))
  BENCH("v_bcnt", (uint32_t)__popc(x) + y)
  BENCH("v_mbcnt_lo", __builtin_amdgcn_mbcnt_lo(y, x))
  BENCH("v_lshl_add", (x << 3) + y)
  BENCH("v_add3", x + y + seed)
  BENCH("v_cvt_f32_i32+back", (uint32_t)(int)((float)(int)x))
  BENCH("v_mul_f32", __float_as_uint(__uint_as_float(x | 0x3f800000u) * 1.0001f))
  BENCH("v_fma_f32", __float_as_uint(__builtin_fmaf(__uint_as_float(x | 0x3f800000u), 1.0001f, 0.5f)))
  BENCH("v_rcp_f32", __float_as_uint(__builtin_amdgcn_rcpf(__uint_as_float(x | 0x3f800000u))))
  BENCH("dpp row_shr add", x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true))
  BENCH("ds_bpermute", (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, (int)x) + 1u)
  BENCH("v_readlane+add", (uint32_t)__builtin_amdgcn_readlane((int)x, 5) + x)
  {   // packed FP32 multiply: 16 independent chains
    f2 a[N];
    for (int i = 0; i < N; i++) a[i] = (f2){1.0f + lane * 1e-3f + i, 1.0f};
    const f2 m = {1.0001f, 0.9999f};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < N; i++) a[i] = a[i] * m;
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < N; i++) s += a[i].x + a[i].y;
    sink ^= __float_as_uint(s);
    if (lane == 0 && blockIdx.x == 0) cyc[nb] = t1 - t0;
    nb++;
  }
  {   // FP64 FMA
    double a[N];
    for (int i = 0; i < N; i++) a[i] = 1.0 + lane * 1e-3 + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < N; i++) a[i] = __builtin_fma(a[i], 0.999, 1e-3);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < N; i++) s += a[i];
    sink ^= (uint32_t)__double2loint(s);
    if (lane == 0 && blockIdx.x == 0) cyc[nb] = t1 - t0;
    nb++;
  }
  out[blockIdx.x * blockDim.x + lane] = sink;
}
int main() {
  const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24", "v_perm_b32", "v_alignbyte",
                         "v_dot4_u32_u8", "v_dot2_u32_u16", "v_sad_u16", "v_pk_max_u16", "v_pk_sub_i16", "v_bcnt+add", "v_mbcnt_lo", "v_lshl_add", "v_add3",
                         "cvt f32<->i32 (2)", "v_mul_f32(+or)", "v_fma_f32(+or)", "v_rcp_f32(+or)", "dpp row_shr add", "ds_bpermute+add", "v_readlane+add",
                         "v_pk_mul_f32", "v_fma_f64"};
  uint32_t* out; long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, iters, 12345u, 77u);
  hipDeviceSynchronize();
  long long h[64];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 26; i++) printf("%-22s %6.2f cycles per instruction (clock64 ticks / %d)\n", names[i], (double)h[i] / (iters * 16.0), iters * 16);
  return 0;
}
