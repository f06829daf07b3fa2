// Micro-benchmark (developer tool, r05): the 24 x 24 diagonal block of ba_solve on ONE wave, lane = row, in two forms:
//   A  as ba_solve r04: multipliers broadcast by v_readlane (2 + 1 instructions per update)
//   B  rows 0..15 in DPP row 0, rows 16..23 in DPP row 1: the update of column k is ONE v_fmac_f64_dpp row_newbcast - the multipliers of
//      rows 0..15 are copied into DPP row 1 once per step (v_permlane16_swap), columns >= 16 broadcast inside DPP row 1
// prints cycles per block and checks that both give the same bits.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast ldlt_diag24.hip -o ldlt_diag24
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#define NB 24
__device__ __forceinline__ double shfl_d(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ double recip_d(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
#define PS_DPP_CASES(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
__device__ __forceinline__ void fmac_row_bcast(double& acc, double src, double mul, int k, bool fresh) {
  switch (k) {
#define PS_OP(K) case K: \
    if (fresh) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul)); \
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul)); \
    break;
    PS_DPP_CASES(PS_OP)
#undef PS_OP
    default: break;
  }
}
// every lane of DPP rows 1 and 3 gets the value of the lane 16 below it (rows 0 and 2 keep theirs)
__device__ __forceinline__ double copy_row_up(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]);
}
template <int MODE>
__global__ void k(const double* in, double* out, long long* cyc) {
  const int lane = threadIdx.x;
  double a[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) a[c] = (lane < NB && c <= lane) ? in[lane * NB + c] : 0.0;
  __syncthreads();
  const long long t0 = clock64();
  double myrd = 0, mydiag = 0;
  if (MODE == 0) {
    double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      myrd = lane == j ? rd : myrd; mydiag = lane == j ? d : mydiag;
      const double l = a[j] * rd;
      double dn = 1.0, rdn = 1.0;
      if (j + 1 < NB) { a[j + 1] -= a[j] * shfl_d(l, j + 1); dn = shfl_d(a[j + 1], j + 1); rdn = recip_d(dn); }
#pragma unroll
      for (int kk = j + 2; kk < NB; kk++) a[kk] -= a[j] * shfl_d(l, kk);
      a[j] = lane > j ? l : a[j];
      d = dn; rd = rdn;
    }
  } else {
    double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      myrd = lane == j ? rd : myrd; mydiag = lane == j ? d : mydiag;
      const double l = a[j] * rd, nl = -l;
      // the multipliers of rows 0..15 where the lanes of DPP row 1 can reach them
      double m = j < 15 ? copy_row_up(nl) : nl;
      asm volatile("s_nop 4" : "+v"(m));
      double dn = 1.0, rdn = 1.0;
      if (j + 1 < NB) {
        if (j + 1 < 16) fmac_row_bcast(a[j + 1], m, a[j], j + 1, true);
        else fmac_row_bcast(a[j + 1], nl, a[j], j + 1 - 16, true);
        dn = shfl_d(a[j + 1], j + 1);
        rdn = recip_d(dn);
      }
#pragma unroll
      for (int kk = j + 2; kk < NB; kk++) {
        if (kk < 16) fmac_row_bcast(a[kk], m, a[j], kk, false);
        else fmac_row_bcast(a[kk], nl, a[j], kk - 16, false);
      }
      a[j] = lane > j ? l : a[j];
      d = dn; rd = rdn;
    }
  }
  const long long t1 = clock64();
  if (lane < NB) {
    for (int c = 0; c < NB; c++) out[lane * (NB + 2) + c] = c < lane ? a[c] : 0.0;     // strictly lower part: the multipliers
    out[lane * (NB + 2) + NB] = mydiag; out[lane * (NB + 2) + NB + 1] = myrd;
  }
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double h[NB * NB];
  for (int r = 0; r < NB; r++) for (int c = 0; c < NB; c++) h[r * NB + c] = r == c ? 30.0 + r : 1.0 / (1 + r + c) + 0.01 * ((r * 7 + c * 3) % 5);
  double *din, *dout; long long* dcyc;
  (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, NB * (NB + 2) * 8); (void)hipMalloc(&dcyc, 64);
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  double res[2][NB * (NB + 2)];
  for (int mode = 0; mode < 2; mode++) {
    long long c = 0;
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, din, dout, dcyc); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, din, dout, dcyc);
      (void)hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    }
    (void)hipMemcpy(res[mode], dout, sizeof(res[0]), hipMemcpyDeviceToHost);
    printf("24 x 24 block, %-60s %6lld cycles, %5.0f per pivot\n", mode == 0 ? "v_readlane broadcasts (ba_solve r04)" : "two DPP rows, v_permlane16_swap + row_newbcast", c, (double)c / NB);
  }
  printf("results %s\n", memcmp(res[0], res[1], sizeof(res[0])) == 0 ? "identical (bitwise)" : "DIFFER");
  {  // host LDL^T in long double: which of the two is the rounding of the true factor?
    static long double A[NB][NB]; long double D[NB];
    for (int r = 0; r < NB; r++) for (int c = 0; c < NB; c++) A[r][c] = h[r * NB + c];
    for (int j = 0; j < NB; j++) {
      long double d = A[j][j]; for (int q = 0; q < j; q++) d -= A[j][q] * A[j][q] * D[q]; D[j] = d;
      for (int i = j + 1; i < NB; i++) { long double v = A[i][j]; for (int q = 0; q < j; q++) v -= A[i][q] * A[j][q] * D[q]; A[i][j] = v / d; }
    }
    double e0 = 0, e1 = 0;
    for (int r = 0; r < NB; r++) { e0 = fmax(e0, fabs((double)(res[0][r * (NB + 2) + NB] - D[r]))); e1 = fmax(e1, fabs((double)(res[1][r * (NB + 2) + NB] - D[r]))); }
    printf("max |pivot - long double pivot|: readlane form %.3g, DPP form %.3g\n", e0, e1);
  }
  if (memcmp(res[0], res[1], sizeof(res[0])) != 0)
    for (int r = 0; r < NB; r++) { printf("row %2d: ", r); for (int c = 0; c < NB + 2; c++) printf("%c", res[0][r * (NB + 2) + c] == res[1][r * (NB + 2) + c] ? '.' : 'X'); printf("\n"); }
  return 0;
}
