// CDNA4 kernel for the pose-only optimisers:
//   Optimizer::PoseOptimization            /root/reference/src/Optimizer.cc:249-477      (mode 0)
//   Optimizer::CFSE3ObjStateOptimization   :479-753                                      (mode 1)
// One persistent workgroup per problem runs the reference's whole schedule on the device — 4 rounds of
// up to 10 Levenberg–Marquardt iterations with up to 10 damping trials each, chi-square re-classification
// between rounds — without a host round trip.  A problem has k SE3 vertices (k = 1 for PoseOptimization,
// k = number of objects for CFSE3: g2o solves them as ONE graph, i.e. one shared lambda / gain ratio /
// stop rule and a block-diagonal 6k x 6k system), each with its own contiguous range of unary edges.
//
// Per LM iteration every thread evaluates its edges (error, analytic 2x6/3x6 Jacobian, Huber weight),
// keeps 21 + 6 + 1 FP64 partial sums in registers (upper triangle of J^T W J, J^T W e, robust chi2),
// reduces them with wave shuffles and one LDS pass in a fixed order, and every thread then solves the
// 6x6 system redundantly in registers (no broadcast, no divergence).
//
// g2o semantics reproduced (Thirdparty/g2o/g2o/...): core/optimization_algorithm_levenberg.cpp:61-189,
// core/base_unary_edge.hpp:43-73, core/robust_kernel_impl.cpp:78-91, types/types_six_dof_expmap.cpp:266-360,
// types/se3quat.h; include/g2o_Object.h:407-422 (translation prior with the numeric Jacobian of
// core/base_unary_edge.hpp:83-121).  Edge errors are cached exactly where g2o caches them: the
// classification reads the chi2 of the LAST computeActiveErrors (possibly a rejected trial).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include "opt_plan.h"
#include "se3.h"

namespace {

#ifndef PO_T
#define PO_T 256
#endif
#ifndef PO_JMAX
#define PO_JMAX 9      // damping trials whose solves are done side by side once a first trial was rejected (g2o stops after 10 trials: 1 + 9)
#endif
static_assert(PO_T % 64 == 0, "only whole waves may leave the kernel early (see NT below)");

struct PoShared {
  double red[PO_T / 64][28];
  double out[28];
  double outg[PO_T / 64][28];       // group_sum: the sums of the vertices worked on side by side
  double chiv[PS_PO_MAX_K];         // robust chi2 per vertex of the last linearisation
  double pose[PS_PO_MAX_K][7];      // current estimates
  double pose0[PS_PO_MAX_K][7];     // estimates at entry (PoseOptimization restarts every round from them)
  double H[PS_PO_MAX_K][21];
  double b[PS_PO_MAX_K][6];
  double cpose[64][7];              // estimates of the damping trials solved ahead (index j k + o: trial j of the batch, vertex o) ...
  double cx[64][6];                 // ... and the increment vector g2o would hold after each of them
  unsigned cok;                     // bit j: every block of trial j factorised
  double prior_obs[PS_PO_MAX_K][3];
  double prior_err[PS_PO_MAX_K][3];
  uint8_t prior_robust[PS_PO_MAX_K];
  int icount;
  double red1[2][PO_T / 64];        // block_sum1: the waves' partial sums, two buffers used in turn
  int nv[PS_PO_MAX_K];              // valid edges per vertex = the length of its compacted edge list
  int wcnt[PO_T / 64];              // compaction: valid slots in each wave's share of a vertex's range
};

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

// One value-halving step of the reductions below between the lanes l and l ^ 32 (W = 32) or l ^ 16 (W = 16): the lower lane of a pair keeps a and
// adds its partner's a, the upper lane keeps b and adds its partner's b.  v_permlane32_swap / v_permlane16_swap exchange the upper half (odd
// rows) of one register with the lower half (even rows) of the other: after the swap a' = [a.lo, b.lo], b' = [a.hi, b.hi], and a' + b' is that sum
// on every lane - two swaps and an add per double where select / select / shuffle / add were seven instructions.  Same operands, same order.
template <int W>
__device__ __forceinline__ double swap_add_d(double a, double b) {
  int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  if (W == 32) {
    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false), hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  } else {
    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false), hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  }
}

// sums acc[0..N) over the workgroup in a fixed order; the result lands in s.out[0..N) for every thread
template <int N>
__device__ void block_sum(double* acc, PoShared& s, int nthr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (N > 4) {
    // Many values: instead of one 6-step butterfly per value (6 N exchanges), every step halves the number of values a lane
    // carries - the lane keeps one half, sends the other to its partner and adds what it receives - so the wave makes
    // 16 + 8 + 4 + 2 + 1 + 1 exchanges for up to 32 values.  After the five halving steps lane l holds value (l >> 1) & 31
    // summed over its 32-lane class; the last step adds the two classes.
    static_assert(N <= 32, "block_sum: at most 32 values");
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = i < N ? acc[i] : 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = swap_add_d<32>(v[i], v[i + 16]);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = swap_add_d<16>(v[i], v[i + 8]);
#pragma unroll
    for (int step = 2; step < 5; step++) {
      const int d = 32 >> step, half = 16 >> step;
      const bool upper = (lane & d) != 0;
#pragma unroll
      for (int i = 0; i < half; i++) {
        const double keep = upper ? v[i + half] : v[i];
        const double send = upper ? v[i] : v[i + half];
        v[i] = keep + shfl_xor_d(send, d);
      }
    }
    const double total = v[0] + shfl_xor_d(v[0], 1);
    __syncthreads();   // protect s.out / s.red from the previous use
    if ((lane & 1) == 0 && (lane >> 1) < N) s.red[wave][lane >> 1] = total;
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) {
      double v = acc[i];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += shfl_xor_d(v, d);
      acc[i] = v;
    }
    __syncthreads();   // protect s.out / s.red from the previous use
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < N; i++) s.red[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double v = 0;
#pragma unroll
    for (int w = 0; w < nthr / 64; w++) v += s.red[w][threadIdx.x];
    s.out[threadIdx.x] = v;
  }
  __syncthreads();
}

// ONE value summed over the workgroup, returned to every thread (r05: the damping trials end in this reduction - 25 to 50 of them per
// call - and it cost as much as the trial's edge pass: six ds_bpermute pairs and three workgroup barriers).  Inside the wave the same
// butterfly (partners 32, 16, 8, 4, 2, 1 lanes away, the same sums in the same order) on the vector ALUs alone - v_permlane32/16_swap,
// then DPP row rotate / shifts / quad permutes -, across the waves ONE barrier: the partial sums go to one of two buffers used in turn
// (`turn`, the same in every thread: a thread that is still reading a buffer cannot be overtaken by the write after next - the
// barrier of the call in between holds it back), and every thread adds them itself, wave 0 first.
__device__ __forceinline__ double wave_sum_d(double v) {
  v = swap_add_d<32>(v, v);
  v = swap_add_d<16>(v, v);
  {   // 8 lanes away: rotate the 16-lane row by 8
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, false);
    v += __hiloint2double(hi, lo);
  }
  {   // 4 lanes away: lanes 0-3 / 8-11 of a row take from 4 above (row_shl:4, banks 0 and 2), lanes 4-7 / 12-15 from 4 below (row_shr:4, banks 1 and 3)
    int lo = __double2loint(v), hi = __double2hiint(v);
    int tl = __builtin_amdgcn_update_dpp(lo, lo, 0x104 /* row_shl:4 */, 0xF, 0x5, false);
    tl = __builtin_amdgcn_update_dpp(tl, lo, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
    int th = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xF, 0x5, false);
    th = __builtin_amdgcn_update_dpp(th, hi, 0x114, 0xF, 0xA, false);
    v += __hiloint2double(th, tl);
  }
  {   // 2 lanes away
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    v += __hiloint2double(hi, lo);
  }
  {   // the neighbour
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    v += __hiloint2double(hi, lo);
  }
  return v;
}
__device__ __forceinline__ double block_sum1(double v, PoShared& s, int nthr, int& turn) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) s.red1[turn][threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
#pragma unroll
  for (int w = 0; w < PO_T / 64; w++) if (w < nthr / 64) t += s.red1[turn][w];
  turn ^= 1;
  return t;
}

// The same reduction per GROUP of waves: the workgroup is split into G = 1, 2 or 4 groups of PO_T / G threads that work on different
// vertices of a CFSE3 graph at the same time (g = the thread's group, tg its index inside it); sums land in s.outg[g][0..N).
// With G = 1 this is block_sum: same order of additions.
template <int N>
__device__ void group_sum(double* acc, PoShared& s, int G, int g, int tg, int nthr) {
  static_assert(N > 4 && N <= 32, "group_sum: 5 .. 32 values");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[32];
#pragma unroll
  for (int i = 0; i < 32; i++) v[i] = i < N ? acc[i] : 0.0;
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = swap_add_d<32>(v[i], v[i + 16]);
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = swap_add_d<16>(v[i], v[i + 8]);
#pragma unroll
  for (int step = 2; step < 5; step++) {
    const int d = 32 >> step, half = 16 >> step;
    const bool upper = (lane & d) != 0;
#pragma unroll
    for (int i = 0; i < half; i++) {
      const double keep = upper ? v[i + half] : v[i];
      const double send = upper ? v[i] : v[i + half];
      v[i] = keep + shfl_xor_d(send, d);
    }
  }
  const double total = v[0] + shfl_xor_d(v[0], 1);
  __syncthreads();   // protect s.outg / s.red from the previous use
  if ((lane & 1) == 0 && (lane >> 1) < N) s.red[wave][lane >> 1] = total;
  __syncthreads();
  const int wpg = (nthr / 64) / G;
  if (tg < N) {
    double sum = 0;
    for (int w = 0; w < wpg; w++) sum += s.red[g * wpg + w][tg];
    s.outg[g][tg] = sum;
  }
  __syncthreads();
}

__device__ __forceinline__ Se3 load_pose(const double* p) {
  Se3 T;
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
  T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
  return T;
}
__device__ __forceinline__ void store_pose(double* p, const Se3& T) {
  p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2];
  p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
}

// RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91): rho[0] and rho[1]
__device__ __forceinline__ void huber(double e, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
  else { const double sq = sqrt(e); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}

// error of a projection edge at pose T (EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose)
__device__ __forceinline__ void edge_error(const Se3& T, const PoProb& P, const float* xw, const float* ob,
                                           bool mono, double p[3], double e[3]) {
  const double X[3] = {(double)xw[0], (double)xw[1], (double)xw[2]};
  se3_map(T, X, p);
  if (mono) {
    e[0] = (double)ob[0] - (p[0] / p[2] * (double)P.fx + (double)P.cx);
    e[1] = (double)ob[1] - (p[1] / p[2] * (double)P.fy + (double)P.cy);
    e[2] = 0.0;
  } else {
    const float invz = (float)(1.0 / p[2]);   // `const float invz = 1.0f/trans_xyz[2]` (types_six_dof_expmap.cpp:301)
    const double u = p[0] * (double)invz * (double)P.fx + (double)P.cx;
    const double v = p[1] * (double)invz * (double)P.fy + (double)P.cy;
    e[0] = (double)ob[0] - u;
    e[1] = (double)ob[1] - v;
    e[2] = (double)ob[2] - (u - (double)P.bf * (double)invz);
  }
}

// state byte per edge: bit0 valid, bit1 level 1 (outlier, inactive), bit2 mono
// one edge's inputs; the edge loops fetch the NEXT edge before they work on the current one, so a pass pays one L2 round trip
// instead of one per edge (a thread owns n / PO_T edges, PO_T apart)
// (r06) The edges a thread walks are the COMPACTED list of its vertex: at entry the valid slots of [e_begin, e_end) are gathered, in slot
// order, into 32-byte records at the front of the range (`cedge`: {Xw, 1 / sigma^2 | observation, slot index}; state and cached chi2 live at the
// same compact positions).  The tracker hands over a frame's ~2000 feature slots of which 400 - 800 carry a map point: a wave stepped
// through every slot it owned and paid an edge's full cost wherever ANY of its 64 lanes had one - eight steps for two to three edges.
struct PoEdge { float x0, x1, x2, o0, o1, o2, is2; uint8_t st; };
__device__ __forceinline__ PoEdge po_load(const float4* cedge, const uint8_t* state, int i) {
  const float4 a = cedge[2 * (size_t)i], b = cedge[2 * (size_t)i + 1];
  PoEdge e;
  e.x0 = a.x; e.x1 = a.y; e.x2 = a.z; e.is2 = a.w;
  e.o0 = b.x; e.o1 = b.y; e.o2 = b.z; e.st = state[i];
  return e;
}
#define ST_VALID 1
#define ST_LVL1 2
#define ST_MONO 4

#ifdef PS_PO_PROFILE   // developer build: 100 MHz ticks per phase of problem 0, printed by the kernel
#define POP_DECL long long po_t0 = wall_clock64(), po_tt = po_t0, po_ph[5] = {0, 0, 0, 0, 0}; int po_n[5] = {0, 0, 0, 0, 0}
#define POP_MARK(k) do { const long long _n = wall_clock64(); po_ph[k] += _n - po_tt; po_tt = _n; po_n[k]++; } while (0)
#define POP_PRINT() do { if (threadIdx.x == 0 && blockIdx.x == 0) printf("pose_lm ticks: setup/other %lld (%d) linearize %lld (%d) solve %lld (%d) trial %lld (%d) classify %lld (%d) total %lld\n", po_ph[0], po_n[0], po_ph[1], po_n[1], po_ph[2], po_n[2], po_ph[3], po_n[3], po_ph[4], po_n[4], wall_clock64() - po_t0); } while (0)
#else
#define POP_DECL
#define POP_MARK(k)
#define POP_PRINT()
#endif
#ifndef PO_OCC_ATTR
#define PO_OCC_ATTR      // developer knob (tools/build_variant.sh): e.g. __attribute__((amdgpu_waves_per_eu(3,3))) - 222 registers give two workgroups per
#endif                   // CU; capped to 168 / 128 the kernel spills and the headline loses 2 % (r04)
__global__ __launch_bounds__(PO_T) PO_OCC_ATTR void pose_lm(const PoProb* probs, const PoVertex* verts, const float* xw,
                                                const float* obs, const float* inv_sigma2, const uint8_t* valid,
                                                uint8_t* outlier, double* chi2c, uint8_t* state, float4* cedge, double* poses,
                                                int32_t* results, double* trace) {
  __shared__ PoShared s;
  POP_DECL;
  const PoProb P = probs[blockIdx.x];
  const int tid = threadIdx.x;
  const int k = P.k;
  // A CFSE3 graph of one or two objects - a few hundred edges - runs on 128 of the workgroup's threads: the other two waves leave here
  // (a barrier only counts the waves that are still alive).  With fewer lanes the reductions and the barriers between the short edge
  // passes weigh less (0.84 -> 0.70 ms per step in the tracker, r04); graphs of more objects keep all four waves (up to four objects
  // are linearised side by side, one per wave), PoseOptimization its 256 lanes for a frame's ~1000 edges.
  const int NT = (P.mode == 1 && k <= 2) ? 128 : PO_T;
  if (tid >= NT) return;
  // CFSE3 graphs: G vertices are linearised side by side, each by a group of GT threads (whole waves)
  const int G = min(k >= 4 ? 4 : (k >= 2 ? 2 : 1), NT / 64), GT = NT / G, g = tid / GT, tg = tid - g * GT;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  double* tr = trace ? trace + (size_t)blockIdx.x * PS_PO_TRACE * 3 : nullptr;
  int ntr = 0;

  // ---- graph construction (Optimizer.cc:262-377 / :506-637) ----
  for (int i = tid; i < k * 7; i += NT) {
    const double v = poses[(size_t)P.v_off * 7 + i];
    (&s.pose[0][0])[i] = v;
    (&s.pose0[0][0])[i] = v;
  }
  if (tid < k) {
    s.prior_robust[tid] = 1;
    for (int c = 0; c < 3; c++) s.prior_obs[tid][c] = poses[(size_t)(P.v_off + tid) * 7 + c];
    // g2o keeps the increment in a zero-initialised vector: when the very first factorisation of a problem fails the
    // update is exp(0) (and rho = -inf: the damping is raised and the trial repeated) - never leftover LDS contents
  }
  // the edge lists: every vertex's valid slots, in slot order, compacted to the front of its range.  Wave w takes the w-th share of the
  // range (whole 64-slot rows), counts its valid slots, and after one barrier writes them behind the shares before it.
  int turn = 0;                                     // block_sum1's buffer in use
  int nInitial = 0;
  {
    const int lane = tid & 63, wave = tid >> 6, nw = NT / 64;
    for (int o = 0; o < k; o++) {
      const PoVertex V = verts[P.v_off + o];
      const int n = V.e_end - V.e_begin;
      const int per = (((n + nw - 1) / nw) + 63) & ~63;
      const int wb = V.e_begin + wave * per, we = min(wb + per, V.e_end);
      int c = 0;
      for (int i0 = wb; i0 < we; i0 += 512) {
        uint8_t f[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 64 + lane; f[u] = i < we ? valid[i] : (uint8_t)0; }
#pragma unroll
        for (int u = 0; u < 8; u++) c += __popcll(__ballot(f[u] != 0));
      }
      if (lane == 0) s.wcnt[wave] = c;
      __syncthreads();
      int run = V.e_begin, tot = 0;
      for (int w = 0; w < nw; w++) { const int cw = s.wcnt[w]; if (w < wave) run += cw; tot += cw; }
      for (int i0 = wb; i0 < we; i0 += 256) {
        uint8_t f[4];
        float e[4][7];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * 64 + lane;
          f[u] = i < we ? valid[i] : (uint8_t)0;
          const int ic = i < we ? i : wb;
          e[u][0] = xw[3 * (size_t)ic]; e[u][1] = xw[3 * (size_t)ic + 1]; e[u][2] = xw[3 * (size_t)ic + 2];
          e[u][3] = obs[3 * (size_t)ic]; e[u][4] = obs[3 * (size_t)ic + 1]; e[u][5] = obs[3 * (size_t)ic + 2];
          e[u][6] = inv_sigma2[ic];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * 64 + lane;
          const unsigned long long bal = __ballot(f[u] != 0);
          if (f[u]) {
            const int pos = run + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
            cedge[2 * (size_t)pos] = make_float4(e[u][0], e[u][1], e[u][2], e[u][6]);
            cedge[2 * (size_t)pos + 1] = make_float4(e[u][3], e[u][4], e[u][5], __int_as_float(i));
            state[pos] = ST_VALID | (e[u][5] < 0.f ? ST_MONO : 0);
            chi2c[pos] = 0.0;
            outlier[i] = 0;          // mvbOutlier[i] = false (Optimizer.cc:299,335)
          }
          run += __popcll(bal);
        }
      }
      if (tid == 0) s.nv[o] = tot;
      nInitial += tot;
      __syncthreads();               // s.wcnt is free again; the records are visible to the whole workgroup
    }
  }
  const int nTotalEdges = nInitial + (P.mode == 1 ? k : 0);
  if (nTotalEdges < 15) {   // Optimizer.cc:376-377 / :638-639
    if (tid == 0) {
      results[blockIdx.x] = 0;
      if (tr) tr[2] = -1;   // empty trace
    }
    return;
  }
  bool robust = true;
  int nBadTotal = 0;
  int last_c = -1;     // where in s.cx the increment vector of the last trial that ran lives (g2o's vector persists over trials, iterations and rounds); -1: still zero

  for (int it = 0; it < 4; it++) {
    if (P.mode == 0) {   // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw)) every round (Optimizer.cc:394)
      __syncthreads();
      for (int i = tid; i < k * 7; i += NT) (&s.pose[0][0])[i] = (&s.pose0[0][0])[i];
      __syncthreads();
    }
    // active set = level-0 edges; a vertex without active edges is not optimised
    int nact = 0;
    for (int o = 0; o < k; o++) {
      const int eb = verts[P.v_off + o].e_begin, ee = eb + s.nv[o];
      for (int i = eb + tid; i < ee; i += NT) nact += ((state[i] & (ST_VALID | ST_LVL1)) == ST_VALID) ? 1 : 0;
    }
    const bool any_active = (int)block_sum1((double)nact, s, NT, turn) > 0 || P.mode == 1;

    if (any_active) {
      double lambda = 0, ni = 2;
      int nBad = 0;
      for (int iter = 0; iter < 10; iter++) {
        // ---- computeActiveErrors + buildSystem ----
        POP_MARK(0);
        double chi_total = 0;
        for (int o0 = 0; o0 < k; o0 += G) {
          // G vertices at a time, each by its group of GT threads (PoseOptimization: one vertex, the whole workgroup)
          const int o = min(o0 + g, k - 1);
          const bool have = o0 + g < k;
          PoVertex V = have ? verts[P.v_off + o] : PoVertex{0, 0};
          if (have) V.e_end = V.e_begin + s.nv[o];
          const Se3 T = load_pose(s.pose[o]);
          double acc[28];
#pragma unroll
          for (int a = 0; a < 28; a++) acc[a] = 0;
          PoEdge nx = {};
          if (V.e_begin + tg < V.e_end) nx = po_load(cedge, state, V.e_begin + tg);
          for (int i = V.e_begin + tg; i < V.e_end; i += GT) {
            const PoEdge ed = nx;
            if (i + GT < V.e_end) nx = po_load(cedge, state, i + GT);
            const uint8_t st = ed.st;
            if ((st & (ST_VALID | ST_LVL1)) != ST_VALID) continue;
            const bool mono = st & ST_MONO;
            double p[3], e[3];
            const float exw[3] = {ed.x0, ed.x1, ed.x2}, eob[3] = {ed.o0, ed.o1, ed.o2};
            edge_error(T, P, exw, eob, mono, p, e);
            const double w = (double)ed.is2;
            const double chi2 = (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) * w;
            chi2c[i] = chi2;
            double rho0 = chi2, rho1 = 1.0;
            if (robust) huber(chi2, mono ? deltaMono : deltaStereo, rho0, rho1);
            acc[27] += rho0;
            // Jacobian (types_six_dof_expmap.cpp:266-292 / :330-360)
            const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
            const double fx = (double)P.fx, fy = (double)P.fy, bf = (double)P.bf;
            double J[3][6];
            J[0][0] = x * y * invz_2 * fx; J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx;
            J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = x * invz_2 * fx;
            J[1][0] = (1 + y * y * invz_2) * fy; J[1][1] = -x * y * invz_2 * fy; J[1][2] = -x * invz * fy;
            J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = y * invz_2 * fy;
            if (mono) {
#pragma unroll
              for (int c = 0; c < 6; c++) J[2][c] = 0;
            } else {
              J[2][0] = J[0][0] - bf * y * invz_2; J[2][1] = J[0][1] + bf * x * invz_2; J[2][2] = J[0][2];
              J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - bf * invz_2;
            }
            const double wo = rho1 * w;
            // J^T W J and J^T W e without the products whose factor is a structural zero of the projection Jacobian (the u and uR rows have
            // no entry in column 4, the v row none in column 3): 18 of the 63 + 3 of the 18 products, and entry (3, 4) altogether.  A product
            // with an exact zero adds nothing to a sum, so the values are the ones the dense expression gives.
            int a = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
#pragma unroll
              for (int c = r; c < 6; c++) {
                const bool u02 = r != 4 && c != 4, u1 = r != 3 && c != 3;      // compile-time after unrolling
                if (u02 && u1) acc[a] += wo * (J[0][r] * J[0][c] + J[1][r] * J[1][c] + J[2][r] * J[2][c]);
                else if (u02) acc[a] += wo * (J[0][r] * J[0][c] + J[2][r] * J[2][c]);
                else if (u1) acc[a] += wo * (J[1][r] * J[1][c]);
                a++;
              }
            }
#pragma unroll
            for (int r = 0; r < 6; r++) {
              if (r == 3) acc[21 + r] -= wo * (J[0][r] * e[0] + J[2][r] * e[2]);
              else if (r == 4) acc[21 + r] -= wo * (J[1][r] * e[1]);
              else acc[21 + r] -= wo * (J[0][r] * e[0] + J[1][r] * e[1] + J[2][r] * e[2]);
            }
          }
          if (P.mode == 1 && tg == 0 && have) {
            // EdgeTransConstraintFromDetction: error = obs - t, information 50 I, Huber(sqrt 5.991) that is never
            // removed, Jacobian by central differences with delta = 1e-9 through oplus (base_unary_edge.hpp:83-121)
            double e[3] = {s.prior_obs[o][0] - T.t[0], s.prior_obs[o][1] - T.t[1], s.prior_obs[o][2] - T.t[2]};
            s.prior_err[o][0] = e[0]; s.prior_err[o][1] = e[1]; s.prior_err[o][2] = e[2];
            const double w = 50.0;
            const double chi2 = (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) * w;
            double rho0 = chi2, rho1 = 1.0;
            if (s.prior_robust[o]) huber(chi2, deltaMono, rho0, rho1);
            acc[27] += rho0;
            double J[3][6];
            const double delta = 1e-9, scalar = 1.0 / (2 * delta);
#pragma unroll
            for (int d = 0; d < 6; d++) {
              double add[6] = {0, 0, 0, 0, 0, 0};
              add[d] = delta;
              const Se3 Tp = se3_mul(se3_exp(add, false), T);
              add[d] = -delta;
              const Se3 Tm = se3_mul(se3_exp(add, false), T);
#pragma unroll
              for (int r = 0; r < 3; r++)
                J[r][d] = scalar * ((s.prior_obs[o][r] - Tp.t[r]) - (s.prior_obs[o][r] - Tm.t[r]));
            }
            const double wo = rho1 * w;
            int a = 0;
            for (int r = 0; r < 6; r++)
              for (int c = r; c < 6; c++) { acc[a] += wo * (J[0][r] * J[0][c] + J[1][r] * J[1][c] + J[2][r] * J[2][c]); a++; }
            for (int r = 0; r < 6; r++) acc[21 + r] -= wo * (J[0][r] * e[0] + J[1][r] * e[1] + J[2][r] * e[2]);
          }
          if (G == 1) {          // PoseOptimization, or a single object: the plain workgroup reduction
            block_sum<28>(acc, s, NT);
            if (tid < 21) s.H[o][tid] = s.out[tid];
            if (tid >= 21 && tid < 27) s.b[o][tid - 21] = s.out[tid];
            if (tid == 27) s.chiv[o] = s.out[27];
          } else {
            group_sum<28>(acc, s, G, g, tg, NT);
            if (have) {
              if (tg < 21) s.H[o][tg] = s.outg[g][tg];
              if (tg >= 21 && tg < 27) s.b[o][tg - 21] = s.outg[g][tg];
              if (tg == 27) s.chiv[o] = s.outg[g][27];
            }
          }
        }
        __syncthreads();
        for (int o = 0; o < k; o++) chi_total += s.chiv[o];      // in vertex order, as the sequential loop added them
        __syncthreads();
        POP_MARK(1);
        double currentChi = chi_total;
        const double iniChi = currentChi;
        if (iter == 0) {   // computeLambdaInit: tau * max |H_jj| over all vertices (levenberg.cpp:166-180)
          double maxDiag = 0;
          for (int o = 0; o < k; o++) {
            const int dg[6] = {0, 6, 11, 15, 18, 20};
            for (int j = 0; j < 6; j++) maxDiag = fmax(fabs(s.H[o][dg[j]]), maxDiag);
          }
          lambda = 1e-5 * maxDiag;
          ni = 2;
          nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        // Solves ahead.  After a rejected trial g2o restores the estimate and retries with lambda *= ni, ni *= 2 on the SAME H and b
        // (levenberg.cpp:139-149), and rejections come in runs (BASELINE config 3: 221 of 1 281 iterations need 2 - 10 trials, 1 486 trials
        // between them; the tracker's frames likewise - profiles/r05_trial_histogram.txt): the whole retry sequence is known when the
        // first trial fails.  The solve is the serial part of a trial (one lane per 6 x 6 block, ~2 us, half of a trial on a frame with
        // 700 edges), so from the second trial on the J trials that are left are solved side by side - lane j k + o of wave 0 takes block o
        // of trial j - and every later trial of the run only takes its estimate out of LDS.  Each trial is still evaluated and decided
        // on its own, in order: same values, same counts.
        int sp_n = 0, sp_i = 0;       // trials solved by the last batch / the next one to use
        do {
          // ---- solve (H + lambda I) x = b per vertex block; x only changes when every block succeeds ----
          bool ok2 = true;
          double scale = 0;
          if (sp_i == sp_n) {
            __syncthreads();      // everybody is done with the batch before (s.cpose / s.cx / s.out of the last trial)
            const int J = qmax == 0 ? 1 : min(min(10 - qmax, PO_JMAX), 64 / k);
            if (tid < 64) {
              const int j = tid / k, o = tid - j * k;
              const bool act = j < J;
              double lam = lambda, nn = ni;
              for (int t = 0; t < j && t < J; t++) { lam *= nn; nn *= 2; }      // what the rejections before trial j will have made of lambda
              bool okv = true;
              double xv[6];
              {
                // unpivoted LDL^T of the 6x6 block (LinearSolverDense uses Eigen::LDLT + isPositive()).  Every loop has
                // compile-time bounds and there is no early exit, so A / D / y live in registers (a `break` or a data-dependent
                // bound sends them to scratch memory, one L2 round trip per access); a failed pivot only clears okv.
                double A[6][6], D[6];
                {
                  int a = 0;
#pragma unroll
                  for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = r; c < 6; c++) { A[r][c] = s.H[o][a]; A[c][r] = s.H[o][a]; a++; }
                }
#pragma unroll
                for (int q = 0; q < 6; q++) A[q][q] += lam;
                // (the 21 divisions by a pivot go through one reciprocal per pivot - hardware estimate + two Newton steps, full
                // double precision - like the LDL^T of the BA solver: a division sequence is a dozen dependent instructions)
                double rD[6];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                  double d = A[q][q];
#pragma unroll
                  for (int u = 0; u < q; u++) d -= A[q][u] * A[q][u] * D[u];
                  if (!(d > 0)) okv = false;
                  D[q] = d;
                  double r = __builtin_amdgcn_rcp(d);
                  double e = __builtin_fma(-d, r, 1.0);
                  r = __builtin_fma(r, e, r);
                  e = __builtin_fma(-d, r, 1.0);
                  rD[q] = __builtin_fma(r, e, r);
#pragma unroll
                  for (int i = q + 1; i < 6; i++) {
                    double v = A[i][q];
#pragma unroll
                    for (int u = 0; u < q; u++) v -= A[i][u] * A[q][u] * D[u];
                    A[i][q] = v * rD[q];
                  }
                }
                double y[6];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                  double v = s.b[o][i];
#pragma unroll
                  for (int u = 0; u < i; u++) v -= A[i][u] * y[u];
                  y[i] = v;
                }
#pragma unroll
                for (int i = 0; i < 6; i++) y[i] *= rD[i];
#pragma unroll
                for (int i = 5; i >= 0; i--) {
                  double v = y[i];
#pragma unroll
                  for (int u = i + 1; u < 6; u++) v -= A[u][i] * xv[u];
                  xv[i] = v;
                }
              }
              // g2o's increment only changes when every block of a trial factorises, and a trial that fails updates with whatever the
              // vector holds: trial j moves by the solution of the last trial <= j that factorised, or by the vector as it was
              const unsigned long long bad = __ballot(act && !okv);
              const unsigned long long vmask = (1ull << k) - 1;
              int src = -1;
              unsigned okmask = 0;
              for (int t = 0; t < J; t++) {
                const bool okt = ((bad >> (t * k)) & vmask) == 0;
                if (okt) okmask |= 1u << t;
                if (okt && t <= j) src = t;
              }
              const int srcl = (src >= 0 && act) ? src * k + o : tid;
              double xj[6], xprev[6];
#pragma unroll
              for (int c = 0; c < 6; c++) xprev[c] = last_c >= 0 ? s.cx[last_c * k + o][c] : 0.0;     // (read before this batch overwrites s.cx: one wave, in order)
#pragma unroll
              for (int c = 0; c < 6; c++) {
                const int lo = __shfl(__double2loint(xv[c]), srcl), hi = __shfl(__double2hiint(xv[c]), srcl);
                xj[c] = src >= 0 ? __hiloint2double(hi, lo) : xprev[c];
              }
              if (act) {
                // update: estimate <- exp(x) * estimate (VertexSE3Expmap::oplusImpl), with whatever x holds
                const Se3 Tn = se3_mul(se3_exp(xj, false), load_pose(s.pose[o]));
                store_pose(s.cpose[tid], Tn);
#pragma unroll
                for (int c = 0; c < 6; c++) s.cx[tid][c] = xj[c];
              }
              if (tid == 0) s.cok = okmask;
            }
            sp_n = J; sp_i = 0;
            __syncthreads();
          }
          // the trial in turn: its estimate and increment stay where the batch left them (s.pose keeps the estimate the trials start from:
          // a rejected trial has nothing to restore)
          const int ci = sp_i * k;
          ok2 = (s.cok >> sp_i) & 1u;
          last_c = sp_i;
          sp_i++;
          POP_MARK(2);
          for (int o = 0; o < k; o++)
            for (int j = 0; j < 6; j++) scale += s.cx[ci + o][j] * (lambda * s.cx[ci + o][j] + s.b[o][j]);
          // ---- computeActiveErrors at the trial estimate ----
          POP_MARK(2);
          double c1[1] = {0};
          for (int o0 = 0; o0 < k; o0 += G) {
            const int o = min(o0 + g, k - 1);
            const bool have = o0 + g < k;
            PoVertex V = have ? verts[P.v_off + o] : PoVertex{0, 0};
            if (have) V.e_end = V.e_begin + s.nv[o];
            const Se3 T = load_pose(s.cpose[ci + o]);
            // four edges per step, all loads issued before the first use: the pass is short (an error and a Huber weight per
            // edge), so the memory round trip would otherwise be paid once per edge
            for (int i0 = V.e_begin + tg; i0 < V.e_end; i0 += 4 * GT) {
              PoEdge eds[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int i = i0 + u * GT;
                eds[u] = po_load(cedge, state, i < V.e_end ? i : i0);
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int i = i0 + u * GT;
                const PoEdge ed = eds[u];
                const uint8_t st = ed.st;
                if (i >= V.e_end || (st & (ST_VALID | ST_LVL1)) != ST_VALID) continue;
                const bool mono = st & ST_MONO;
                double p[3], e[3];
                const float exw[3] = {ed.x0, ed.x1, ed.x2}, eob[3] = {ed.o0, ed.o1, ed.o2};
                edge_error(T, P, exw, eob, mono, p, e);
                const double chi2 = (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) * (double)ed.is2;
                chi2c[i] = chi2;
                double rho0 = chi2, rho1;
                if (robust) huber(chi2, mono ? deltaMono : deltaStereo, rho0, rho1);
                c1[0] += rho0;
              }
            }
            if (P.mode == 1 && tg == 0 && have) {
              const double e0 = s.prior_obs[o][0] - T.t[0], e1 = s.prior_obs[o][1] - T.t[1], e2 = s.prior_obs[o][2] - T.t[2];
              const double chi2 = (e0 * e0 + e1 * e1 + e2 * e2) * 50.0;
              double rho0 = chi2, rho1;
              if (s.prior_robust[o]) huber(chi2, deltaMono, rho0, rho1);
              c1[0] += rho0;
            }
          }
          double tempChi = block_sum1(c1[0], s, NT, turn);
          POP_MARK(3);
          if (!ok2) tempChi = DBL_MAX;
          rho = (currentChi - tempChi) / (scale + 1e-3);
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - se3_cube(2 * rho - 1);
            alpha = fmin(alpha, 2. / 3.);
            lambda *= fmax(1. / 3., alpha);
            ni = 2;
            currentChi = tempChi;
            __syncthreads();
            if (tid < k * 7) (&s.pose[0][0])[tid] = (&s.cpose[ci][0])[tid];       // the accepted estimate
            __syncthreads();
          } else {
            lambda *= ni;
            ni *= 2;
          }
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (tr && tid == 0 && ntr < PS_PO_TRACE) { tr[3 * ntr] = currentChi; tr[3 * ntr + 1] = lambda; tr[3 * ntr + 2] = qmax; }
        ntr++;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;
      }
    }
    // ---- chi-square classification (Optimizer.cc:404-466 / :652-725) ----
    __syncthreads();
    POP_MARK(0);
    int bad = 0;
    for (int o = 0; o < k; o++) {
      const int eb = verts[P.v_off + o].e_begin, ee = eb + s.nv[o];
      const Se3 T = load_pose(s.pose[o]);
      for (int i = eb + tid; i < ee; i += NT) {
        const PoEdge ed = po_load(cedge, state, i);
        const int slot = __float_as_int(cedge[2 * (size_t)i + 1].w);
        uint8_t st = ed.st;
        const bool mono = st & ST_MONO;
        if (st & ST_LVL1) {   // e->computeError() for edges that sat out the round (mvbOutlier[slot] is set exactly when the edge is at level 1)
          double p[3], e[3];
          const float exw[3] = {ed.x0, ed.x1, ed.x2}, eob[3] = {ed.o0, ed.o1, ed.o2};
          edge_error(T, P, exw, eob, mono, p, e);
          chi2c[i] = (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) * (double)ed.is2;
        }
        const float chi2 = (float)chi2c[i];
        if (chi2 > (mono ? 5.991f : 7.815f)) { outlier[slot] = 1; st |= ST_LVL1; bad++; }
        else { outlier[slot] = 0; st &= ~ST_LVL1; }
        state[i] = st;
      }
    }
    if (it == 2) robust = false;   // e->setRobustKernel(0) on the projection edges; the prior keeps its kernel
    nBadTotal = (int)block_sum1((double)bad, s, NT, turn);
    POP_MARK(4);
  }
  __syncthreads();
  for (int i = tid; i < k * 7; i += NT) poses[(size_t)P.v_off * 7 + i] = (&s.pose[0][0])[i];
  if (tid == 0) {
    results[blockIdx.x] = P.mode == 0 ? nInitial - nBadTotal : 1;
    if (tr && ntr < PS_PO_TRACE) tr[3 * ntr + 2] = -1;   // terminator
  }
  POP_PRINT();
}

}  // namespace

extern "C" void psk_pose_lm_launch(const PoProb* probs, int nprob, const PoVertex* verts, const float* xw,
                                   const float* obs, const float* inv_sigma2, const uint8_t* valid, uint8_t* outlier,
                                   double* chi2c, uint8_t* state, void* cedge, double* poses, int32_t* results, double* trace,
                                   hipStream_t st) {
  // cedge: 32 bytes of scratch per edge slot (the compacted edge records), 16-byte aligned
  hipLaunchKernelGGL(pose_lm, dim3(nprob), dim3(PO_T), 0, st, probs, verts, xw, obs, inv_sigma2, valid, outlier, chi2c,
                     state, (float4*)cedge, poses, results, trace);
}
