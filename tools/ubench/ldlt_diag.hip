// Micro-benchmark (developer tool): the lane-per-row LDL^T of an NB x NB diagonal block by ONE wave, as ba_solve's diag_block does it
// (rows in registers, multipliers broadcast by v_readlane, reciprocal = v_rcp_f64 + 2 Newton steps) - shader cycles per block and
// per pivot for a few formulations.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast ldlt_diag.hip -o ldlt_diag
#include <hip/hip_runtime.h>
#include <cstdio>
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
__device__ __forceinline__ double recip1_d(double d) {   // one Newton step
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
template <int NB, int MODE>
__global__ void k(const double* in, double* out, long long* cyc) {
  const int lane = threadIdx.x;
  __shared__ double rdj[64];
  double a[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) a[c] = (lane < NB && c <= lane) ? in[lane * NB + c] : 0.0;
  __syncthreads();
  const long long t0 = clock64();
  if (MODE == 0) {          // as in ba_solve
    double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (lane == j) rdj[j] = rd;
      const double l = lane > j ? a[j] * rd : 0.0;
      const double ld = l * d;
      double dn = 1.0, rdn = 1.0;
      if (j + 1 < NB) { a[j + 1] -= ld * shfl_d(l, j + 1); dn = shfl_d(a[j + 1], j + 1); rdn = recip_d(dn); }
#pragma unroll
      for (int kk = j + 2; kk < NB; kk++) a[kk] -= ld * shfl_d(l, kk);
      if (lane > j) a[j] = l;
      d = dn; rd = rdn;
    }
  } else if (MODE == 1) {   // no LDS store, no lane masks on the multiplier (rows <= j carry garbage that nobody reads)
    double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const double l = a[j] * rd;
      double dn = 1.0, rdn = 1.0;
      if (j + 1 < NB) { a[j + 1] -= a[j] * shfl_d(l, j + 1); dn = shfl_d(a[j + 1], j + 1); rdn = recip_d(dn); }
#pragma unroll
      for (int kk = j + 2; kk < NB; kk++) a[kk] -= a[j] * shfl_d(l, kk);
      a[j] = lane > j ? l : a[j];
      d = dn; rd = rdn;
    }
    if (lane < NB) rdj[lane] = rd + d;
  } else if (MODE == 2) {   // MODE 1 with one Newton step
    double d = shfl_d(a[0], 0), rd = recip1_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const double l = a[j] * rd;
      double dn = 1.0, rdn = 1.0;
      if (j + 1 < NB) { a[j + 1] -= a[j] * shfl_d(l, j + 1); dn = shfl_d(a[j + 1], j + 1); rdn = recip1_d(dn); }
#pragma unroll
      for (int kk = j + 2; kk < NB; kk++) a[kk] -= a[j] * shfl_d(l, kk);
      a[j] = lane > j ? l : a[j];
      d = dn; rd = rdn;
    }
    if (lane < NB) rdj[lane] = rd + d;
  } else {                  // only the pivot chain (no updates of the columns behind the next pivot): the lower bound of this form
    double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const double l = a[j] * rd;
      double dn = 1.0, rdn = 1.0;
      if (j + 1 < NB) { a[j + 1] -= a[j] * shfl_d(l, j + 1); dn = shfl_d(a[j + 1], j + 1); rdn = recip_d(dn); }
      a[j] = lane > j ? l : a[j];
      d = dn; rd = rdn;
    }
    if (lane < NB) rdj[lane] = rd + d;
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < NB; c++) s += a[c];
  out[lane] = s + rdj[lane & 15];
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int NB, int MODE> void run(const double* din, double* dout, long long* dcyc, const char* name) {
  hipLaunchKernelGGL((k<NB, MODE>), dim3(1), dim3(64), 0, 0, din, dout, dcyc);
  hipLaunchKernelGGL((k<NB, MODE>), dim3(1), dim3(64), 0, 0, din, dout, dcyc);
  long long h = 0;
  (void)hipMemcpy(&h, dcyc, 8, hipMemcpyDeviceToHost);
  printf("NB = %2d  %-60s %6lld cycles per block, %5.0f per pivot\n", NB, name, h, (double)h / NB);
}
int main() {
  double h[24 * 24];
  for (int r = 0; r < 24; r++) for (int c = 0; c < 24; c++) h[r * 24 + c] = r == c ? 30.0 + r : 1.0 / (1 + r + c);
  double *din, *dout; long long* dcyc;
  (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, 64 * 8); (void)hipMalloc(&dcyc, 64);
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  run<16, 0>(din, dout, dcyc, "as in ba_solve");
  run<16, 1>(din, dout, dcyc, "no LDS store, no lane masks");
  run<16, 2>(din, dout, dcyc, "... and one Newton step");
  run<16, 3>(din, dout, dcyc, "pivot chain only");
  run<24, 0>(din, dout, dcyc, "as in ba_solve");
  run<24, 1>(din, dout, dcyc, "no LDS store, no lane masks");
  run<24, 3>(din, dout, dcyc, "pivot chain only");
  run<8, 0>(din, dout, dcyc, "as in ba_solve");
  run<8, 3>(din, dout, dcyc, "pivot chain only");
  return 0;
}
