// Reprojection test of Tracking::DynamicStaticDiscrimination (/root/reference/src/Tracking.cc:2099-2181).
// One workgroup per tracked detection: a thread per object point evaluates the FP64 chi-square of "this point did not move"
// (Pc = Trel * (Tco_last * Po), Trel = Tcw_cur * Tcw_last^-1 prepared on the host), the two lists (monocular / stereo) are
// sorted in LDS, everything above 5 x median is dropped and the remainder is summed IN SORTED ORDER by one lane, exactly like
// the reference's std::sort + std::accumulate, so the averages are bit-identical (this file is built without FMA contraction).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "se3.h"
#include "dyn_plan.h"

namespace {
__device__ void bitonic_sort(double* a, int n2, int tid, int nthreads) {
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const double x = a[i], y = a[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[ixj] = x; }
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(256) void dyn_discriminate(const DynProb* probs, const uint8_t* valid, const double* po, const float* obs,
                                                       const float* inv_sigma2, double* out_avg, int32_t* out_n) {
  __shared__ double lst[2][PS_DYN_MAX];
  __shared__ int cnt[2];
  const DynProb P = probs[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid < 2) cnt[tid] = 0;
  __syncthreads();
  for (int j = tid; j < P.n; j += 256) {
    const int g = P.off + j;
    if (!valid[g]) continue;
    double Plc[3], Pc[3];
    se3_map(P.tco, po + 3 * (size_t)g, Plc);
    se3_map(P.trel, Plc, Pc);
    const double invz = 1.0 / Pc[2];
    const double s = (double)inv_sigma2[g];
    const double z0 = P.cx + Pc[0] * invz * P.fx, z1 = P.cy + Pc[1] * invz * P.fy;
    const double e0 = (double)obs[3 * (size_t)g] - z0, e1 = (double)obs[3 * (size_t)g + 1] - z1;
    const float ur = obs[3 * (size_t)g + 2];
    if (ur < 0) {
      lst[0][atomicAdd(&cnt[0], 1)] = e0 * (s * e0) + e1 * (s * e1);
    } else {
      const double z2 = z0 - (double)P.mbf * invz;
      const double e2 = (double)ur - z2;
      lst[1][atomicAdd(&cnt[1], 1)] = e0 * (s * e0) + e1 * (s * e1) + e2 * (s * e2);
    }
  }
  __syncthreads();
  for (int kind = 0; kind < 2; kind++) {
    const int num = cnt[kind];
    int n2 = 1;
    while (n2 < num) n2 <<= 1;
    for (int i = num + tid; i < n2; i += 256) lst[kind][i] = __builtin_huge_val();
    __syncthreads();
    if (num >= 5) bitonic_sort(lst[kind], n2, tid, 256);
  }
  if (tid < 2) {
    const int num = cnt[tid];
    double avg = 0;
    int kept = num;
    if (num >= 5) {
      const double* v = lst[tid];
      const double median = v[num / 2];          // int(size / 2 + 0.5) with integer size / 2
      const double cut = 5 * median;
      kept = 0;
      double sum = 0.0;
      for (int i = 0; i < num; i++)
        if (!(v[i] > cut)) { sum += v[i]; kept++; }
      avg = sum / kept;
    }
    out_avg[2 * blockIdx.x + tid] = avg;
    out_n[2 * blockIdx.x + tid] = kept;
  }
}
}  // namespace

extern "C" void psk_dyn_launch(const DynProb* probs, int nprob, const uint8_t* valid, const double* po, const float* obs,
                               const float* inv_sigma2, double* out_avg, int32_t* out_n, hipStream_t st) {
  hipLaunchKernelGGL(dyn_discriminate, dim3(nprob), dim3(256), 0, st, probs, valid, po, obs, inv_sigma2, out_avg, out_n);
}
