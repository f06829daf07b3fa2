// CDNA4 kernels of the object bundle adjustment — Optimizer::ObjectLocalBundleAdjustment
// (/root/reference/src/Optimizer.cc:755-1075): Schur-complement Levenberg–Marquardt over object keyframe
// poses (g2o::VertexSE3Fix / fixed VertexSE3Expmap) and object-frame points (VertexSBAPointXYZ, marginalised)
// with EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ edges.
//
// The LM state machine of every problem lives in device memory (BaState); the host only enqueues the same
// kernel sequence ("global step") until every problem reports DONE.  Work is spread over the whole chip:
//   ba_begin        per problem : stage entry — chi2/depth classification, active sets, compact pose indices
//   ba_lin_pose     wave per pose : errors, Jacobians, Huber weights; H_pp, b_p, robust chi2, W = J_p^T w J_x
//   ba_lin_point    thread per point : H_ll, b_l
//   ba_post_lin     per problem : chi2 total, lambda init (tau * max diag)
//   ba_prep         thread per point : (H_ll + lambda I)^-1 ; wave per pose : b_s = b_p - sum W D^-1 b_l
//   ba_schur        tile per 8x8 pose blocks : S = H_pp + lambda I - (W D^-1) W^T, 48x48 FP64 tiles via LDS
//   ba_solve        per problem : blocked unpivoted LDL^T of S, forward/backward substitution
//   ba_update       thread per point / pose : x_l = D^-1 (b_l - W^T x_p), oplus, backups, gain-ratio scale
//   ba_error        thread per edge : errors at the trial estimate, robust chi2
//   ba_decide       per problem : gain ratio, lambda update, accept / restore, stop rules, stage changes
// g2o semantics reproduced (Thirdparty/g2o/g2o/...): core/optimization_algorithm_levenberg.cpp:61-189 (LM control),
// core/block_solver.hpp:354-485,502-608 (Schur, back-substitution, setLambda), core/base_binary_edge.hpp:55-120,
// core/robust_kernel_impl.cpp:78-91, core/sparse_optimizer.cpp:61-114,206-266,354-435, types/types_six_dof_expmap.cpp:
// 103-232, types/types_sba.h:40-57, src/g2o_Object.cc:26-56,190-213.  All sums are FP64 and deterministic
// (fixed-order tree reductions, no atomics).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include "ba_plan.h"
#include "se3.h"



namespace {

#define ES_LVL1 1
#define ES_MONO 2

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}
// broadcast lane `src` (wave-uniform index) of a double: v_readlane_b32 x 2, a few cycles — not the LDS-routed
// ds_bpermute a general __shfl costs
__device__ __forceinline__ double shfl_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// 1 / d to full double precision from the hardware estimate and two Newton steps: a third of the latency of the IEEE
// division sequence, and the pivots' reciprocals are reused by every row of the panel
// lane g (0..3, a constant after unrolling) of every DPP quad to the four lanes of the quad
__device__ __forceinline__ double quad_bcast_d(double v, int g) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  switch (g) {
    case 0: return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x00, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x00, 0xF, 0xF, true));
    case 1: return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x55, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x55, 0xF, 0xF, true));
    case 2: return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xAA, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xAA, 0xF, 0xF, true));
    default: return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xFF, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xFF, 0xF, 0xF, true));
  }
}
__device__ __forceinline__ double recip_d(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
// Broadcasts inside a 16-lane DPP row for FP64 (gfx90a+: 64-bit DPP with row_newbcast).  acc += row_lane_k(src) * mul is ONE instruction
// where v_readlane x 2 -> SGPR pair -> v_fma were three; k is a compile-time constant after unrolling (the switch folds).  A DPP operand
// that a VALU instruction wrote one or two issue slots earlier needs two wait states, which the compiler cannot see inside an asm
// statement: `fresh` puts an s_nop 1 in front (the asm statements are volatile, so later ones stay behind the first).
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
__device__ __forceinline__ double row_bcast_d(double v, int k) {   // every lane of a row gets the row's lane k (v may just have been written)
  double r = 0.0;
  switch (k) {
#define PS_OP(K) case K: asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); break;
    PS_DPP_CASES(PS_OP)
#undef PS_OP
    default: break;
  }
  return r;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += shfl_xor_d(v, d);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmax(v, shfl_xor_d(v, d));
  return v;
}
__device__ __forceinline__ Se3 load_pose(const double* p) {
  Se3 T;
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
  return T;
}
__device__ __forceinline__ void store_pose(double* p, const Se3& T) {
  p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2]; p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
}
__device__ __forceinline__ void huber(double e, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
  else { const double sq = sqrt(e); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}
#define DELTA_MONO ((double)(float)2.4476519361420544)    /* (float)sqrt(5.991) */
#define DELTA_STEREO ((double)(float)2.7955321496988727)  /* (float)sqrt(7.815) */

// EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ::computeError (types_six_dof_expmap.h:87-139, .cpp:141-158)
// chi2 of an edge from its error: ONE statement with the operation order spelled out, used by both halves of the linearisation and by the
// trial's error pass - the pose-major and the point-major half must give an edge the same Huber weight whatever the compiler contracts around them
__device__ __forceinline__ double ba_edge_chi2(const double er[3], double w) {
  return __builtin_fma(er[2], er[2], __builtin_fma(er[1], er[1], er[0] * er[0])) * w;
}
__device__ __forceinline__ void ba_error(const Se3& T, const BaProb& P, const double X[3], const float* ob, bool mono,
                                         double p[3], double e[3]) {
  se3_map(T, X, p);
  if (mono) {
    e[0] = (double)ob[0] - (p[0] / p[2] * (double)P.fx + (double)P.cx);
    e[1] = (double)ob[1] - (p[1] / p[2] * (double)P.fy + (double)P.cy);
    e[2] = 0.0;
  } else {
    const float invz = (float)(1.0 / p[2]);
    const double u = p[0] * (double)invz * (double)P.fx + (double)P.cx, v = p[1] * (double)invz * (double)P.fy + (double)P.cy;
    e[0] = (double)ob[0] - u; e[1] = (double)ob[1] - v; e[2] = (double)ob[2] - (u - (double)P.bf * (double)invz);
  }
}
// linearizeOplus of the two binary edges (types_six_dof_expmap.cpp:103-139, :188-232)
__device__ __forceinline__ void ba_jacobians(const double R[9], const BaProb& P, const double p[3], bool mono,
                                             double Jp[3][6], double Jx[3][3]) {
  const double x = p[0], y = p[1], z = p[2], z_2 = z * z;
  const double fx = (double)P.fx, fy = (double)P.fy, bf = (double)P.bf;
  if (mono) {
    const double t0 = fx, t2 = -x / z * fx, t4 = fy, t5 = -y / z * fy;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Jx[0][c] = -1. / z * (t0 * R[c] + t2 * R[6 + c]);
      Jx[1][c] = -1. / z * (t4 * R[3 + c] + t5 * R[6 + c]);
      Jx[2][c] = 0;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Jx[0][c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
      Jx[1][c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
      Jx[2][c] = Jx[0][c] - bf * R[6 + c] / z_2;
    }
  }
  Jp[0][0] = x * y / z_2 * fx; Jp[0][1] = -(1 + (x * x / z_2)) * fx; Jp[0][2] = y / z * fx;
  Jp[0][3] = -1. / z * fx; Jp[0][4] = 0; Jp[0][5] = x / z_2 * fx;
  Jp[1][0] = (1 + y * y / z_2) * fy; Jp[1][1] = -x * y / z_2 * fy; Jp[1][2] = -x / z * fy;
  Jp[1][3] = 0; Jp[1][4] = -1. / z * fy; Jp[1][5] = y / z_2 * fy;
  if (mono) {
#pragma unroll
    for (int c = 0; c < 6; c++) Jp[2][c] = 0;
  } else {
    Jp[2][0] = Jp[0][0] - bf * y / z_2; Jp[2][1] = Jp[0][1] + bf * x / z_2; Jp[2][2] = Jp[0][2];
    Jp[2][3] = Jp[0][3]; Jp[2][4] = 0; Jp[2][5] = Jp[0][5] - bf / z_2;
  }
}
__device__ __forceinline__ bool inv3(const double M[9], double O[9]) {
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  const double id = 1.0 / det;
  O[0] = c00 * id; O[1] = (M[2] * M[7] - M[1] * M[8]) * id; O[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  O[3] = c01 * id; O[4] = (M[0] * M[8] - M[2] * M[6]) * id; O[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  O[6] = c02 * id; O[7] = (M[1] * M[6] - M[0] * M[7]) * id; O[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  return det != 0;
}

// -------------------------------------------------------------------------------------------------------
// Stage entry.  stage 0: all edges level 0, Huber on, 5 iterations (Optimizer.cc:955-957).
// stage 1: chi2 > 5.991|7.815 or depth <= 0 -> level 1, Huber off, 10 iterations (:959-986).
// stage 2: the same test fills the erase list (:988-1012) and the problem is DONE.
// -------------------------------------------------------------------------------------------------------
__device__ void ba_stage_entry(const BaArrays& A, int prob) {
  const BaProb P = A.prob[prob];
  BaState& S = A.state[prob];
  const int tid = threadIdx.x;
  __shared__ int s_stage;
  for (;;) {
    __syncthreads();
    if (S.phase != BA_PH_BEGIN) return;
    const int stage = S.stage;
    if (stage >= 1) {   // classification / erase list on the cached chi2 and the current estimate
      for (int e = tid; e < P.ne; e += 256) {
        const int ge = P.edge_base + e;
        const bool mono = A.e_state[ge] & ES_MONO;
        const Se3 T = load_pose(A.poses + (size_t)(P.pose_base + A.e_pose[ge]) * 7);
        const double* X = A.points + (size_t)(P.point_base + A.e_point[ge]) * 3;
        double p[3];
        se3_map(T, X, p);
        const bool bad = A.chi2c[ge] > (mono ? 5.991 : 7.815) || !(p[2] > 0.0);
        if (stage == 1) { if (bad) A.e_state[ge] |= ES_LVL1; }
        else A.erase[ge] = bad ? 1 : 0;
      }
    }
    __syncthreads();
    if (stage == 2) {
      if (tid == 0) { S.phase = BA_PH_DONE; atomicAdd(A.ndone, 1); }
      return;
    }
    // active sets: a pose / point takes part iff it has a level-0 edge (and the pose is not fixed).  Edge-parallel: every level-0
    // edge marks its two vertices (the per-vertex walks over the CSR lists this replaces were 300 dependent loads per pose; with the
    // single-thread compaction behind them the stage entry took 175 us)
    for (int i = tid; i < P.np; i += 256) A.pidx[P.pose_base + i] = -1;
    for (int l = tid; l < P.nl; l += 256) {
      A.lact[P.point_base + l] = 0;
      A.xl[(size_t)(P.point_base + l) * 3] = 0; A.xl[(size_t)(P.point_base + l) * 3 + 1] = 0; A.xl[(size_t)(P.point_base + l) * 3 + 2] = 0;
    }
    for (int i = tid; i < P.np * 6; i += 256) A.xp[(size_t)P.pose_base * 6 + i] = 0;
    __syncthreads();
    for (int e = tid; e < P.ne; e += 256) {
      const int ge = P.edge_base + e;
      if (A.e_state[ge] & ES_LVL1) continue;
      A.pidx[P.pose_base + A.e_pose[ge]] = 0;      // (every writer stores the same value)
      A.lact[P.point_base + A.e_point[ge]] = 1;
    }
    __syncthreads();
    // compact indices of the active free poses in pose order, count of the active points
    {
      __shared__ int s_cnt[4], s_base, s_nla[4];
      if (tid == 0) s_base = 0;
      int nla = 0;
      for (int l = tid; l < P.nl; l += 256) nla += A.lact[P.point_base + l];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) nla += __shfl_xor(nla, d);
      if ((tid & 63) == 0) s_nla[tid >> 6] = nla;
      __syncthreads();
      for (int i0 = 0; i0 < P.np; i0 += 256) {
        const int i = i0 + tid;
        const bool act = i < P.np && A.pidx[P.pose_base + i] == 0 && !(A.pose_flags[P.pose_base + i] & 1);
        const unsigned long long b = __ballot(act);
        if ((tid & 63) == 0) s_cnt[tid >> 6] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < (tid >> 6); w++) off += s_cnt[w];
        const int a = off + __popcll(b & ((1ull << (tid & 63)) - 1ull));
        if (i < P.np) A.pidx[P.pose_base + i] = act ? a : -1;
        if (act) A.pact[P.pose_base + a] = i;
        __syncthreads();
        if (tid == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
      }
      if (tid == 0) {
        const int npa = s_base, nlat = s_nla[0] + s_nla[1] + s_nla[2] + s_nla[3];
        S.npa = npa; S.nla = nlat;
        S.robust = stage == 0 ? 1 : 0;
        S.iter = 0; S.max_iter = stage == 0 ? 5 : 10; S.trial = 0; S.n_bad = 0;
        if (npa + nlat == 0) S.stage = stage + 1;   // "0 vertices to optimize": optimize() returns without touching anything
        else S.phase = BA_PH_LINEARIZE;
        s_stage = S.stage;
      }
    }
    __syncthreads();
    if (S.phase != BA_PH_BEGIN) return;
    (void)s_stage;
  }
}

// the first global step's stage entry; later ones run at the end of ba_decide, in the workgroup that made the stage change
__global__ __launch_bounds__(256) void ba_begin(BaArrays A) { ba_stage_entry(A, blockIdx.x); }

// -------------------------------------------------------------------------------------------------------
// computeActiveErrors + buildSystem, pose-major: one wave per pose, lanes over the pose's edges.
// -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ba_lin_pose(const BaArrays& A, int bx) {
  const BaProb P = A.prob[blockIdx.y];
  const BaState& S = A.state[blockIdx.y];
  if (S.phase != BA_PH_LINEARIZE) return;
  const int i = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= P.np) return;
  const Se3 T = load_pose(A.poses + (size_t)(P.pose_base + i) * 7);
  double R[9];
  se3_quat_to_R(T.q, R);
  const bool pose_active = A.pidx[P.pose_base + i] >= 0;
  const bool robust = S.robust != 0;
  double acc[28];
#pragma unroll
  for (int a = 0; a < 28; a++) acc[a] = 0;
  const int b = A.csr_off[P.csr_pose_base + i], e = A.csr_off[P.csr_pose_base + i + 1];
  for (int k = b + lane; k < e; k += 64) {
    const int le = A.csr_edges[P.csr_pose_edges_base + k], ge = P.edge_base + le;
    const uint8_t st = A.e_state[ge];
    const int l = A.e_point[ge];
    double* Wb = A.W + P.W_base + ((size_t)i * P.nl + l) * 18;
    if (st & ES_LVL1) {
#pragma unroll
      for (int q = 0; q < 18; q++) Wb[q] = 0;
      continue;
    }
    const bool mono = st & ES_MONO;
    const double* X = A.points + (size_t)(P.point_base + l) * 3;
    double p[3], er[3];
    ba_error(T, P, X, A.e_obs + (size_t)ge * 3, mono, p, er);
    const double w = (double)A.e_is2[ge];
    const double chi2 = ba_edge_chi2(er, w);
    A.chi2c[ge] = chi2;
    double rho0 = chi2, rho1 = 1.0;
    if (robust) huber(chi2, mono ? DELTA_MONO : DELTA_STEREO, rho0, rho1);
    acc[27] += rho0;
    double Jp[3][6], Jx[3][3];
    ba_jacobians(R, P, p, mono, Jp, Jx);
    const double wo = rho1 * w;
    if (pose_active) {
      int a = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) { acc[a] += wo * (Jp[0][r] * Jp[0][c] + Jp[1][r] * Jp[1][c] + Jp[2][r] * Jp[2][c]); a++; }
#pragma unroll
      for (int r = 0; r < 6; r++) acc[21 + r] -= wo * (Jp[0][r] * er[0] + Jp[1][r] * er[1] + Jp[2][r] * er[2]);
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Wb[r * 3 + c] = wo * (Jp[0][r] * Jx[0][c] + Jp[1][r] * Jx[1][c] + Jp[2][r] * Jx[2][c]);
    } else {
#pragma unroll
      for (int q = 0; q < 18; q++) Wb[q] = 0;
    }
  }
#pragma unroll
  for (int a = 0; a < 28; a++) acc[a] = wave_sum(acc[a]);
  if (lane == 0) {
    double* H = A.Hpp + (size_t)(P.pose_base + i) * 36;
    int a = 0;
    for (int r = 0; r < 6; r++)
      for (int c = r; c < 6; c++) { H[r * 6 + c] = acc[a]; H[c * 6 + r] = acc[a]; a++; }
    for (int r = 0; r < 6; r++) A.bp[(size_t)(P.pose_base + i) * 6 + r] = acc[21 + r];
    A.part[P.part_base + i] = acc[27];
  }
}

// point-major half of buildSystem: H_ll and b_l.  16 lanes share one point (its edges are strided over them and the
// 6 + 3 sums are combined with a fixed-order butterfly over the 16 lanes); the error cache is already fresh.
__device__ __forceinline__ void ba_jx(const double R[9], const BaProb& P, const double p[3], bool mono, double Jx[3][3]) {
  const double x = p[0], y = p[1], z = p[2], z_2 = z * z;
  const double fx = (double)P.fx, fy = (double)P.fy, bf = (double)P.bf;
  if (mono) {
    const double t0 = fx, t2 = -x / z * fx, t4 = fy, t5 = -y / z * fy;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Jx[0][c] = -1. / z * (t0 * R[c] + t2 * R[6 + c]);
      Jx[1][c] = -1. / z * (t4 * R[3 + c] + t5 * R[6 + c]);
      Jx[2][c] = 0;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Jx[0][c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
      Jx[1][c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
      Jx[2][c] = Jx[0][c] - bf * R[6 + c] / z_2;
    }
  }
}
__device__ __forceinline__ void ba_lin_point(const BaArrays& A, int bx) {
  const BaProb P = A.prob[blockIdx.y];
  const BaState& S = A.state[blockIdx.y];
  if (S.phase != BA_PH_LINEARIZE) return;
  const int l = bx * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  const bool lv = l < P.nl;
  const bool robust = S.robust != 0;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // H00 H01 H02 H11 H12 H22 b0 b1 b2
  if (lv) {
    const double* X = A.points + (size_t)(P.point_base + l) * 3;
    const int b = A.csr_off[P.csr_point_base + l], e = A.csr_off[P.csr_point_base + l + 1];
    for (int k = b + sub; k < e; k += 16) {
      const int ge = P.edge_base + A.csr_edges[P.csr_point_edges_base + k];
      const uint8_t st = A.e_state[ge];
      if (st & ES_LVL1) continue;
      const bool mono = st & ES_MONO;
      const Se3 T = load_pose(A.poses + (size_t)(P.pose_base + A.e_pose[ge]) * 7);
      double R[9], p[3], er[3], Jx[3][3];
      se3_quat_to_R(T.q, R);
      ba_error(T, P, X, A.e_obs + (size_t)ge * 3, mono, p, er);
      ba_jx(R, P, p, mono, Jx);
      const double w = (double)A.e_is2[ge];
      double rho0, rho1 = 1.0;
      // (the edge's chi2 evaluated here as ba_lin_pose evaluates it: the two halves run in one launch, the cache is not written yet)
      if (robust) huber(ba_edge_chi2(er, w), mono ? DELTA_MONO : DELTA_STEREO, rho0, rho1);
      const double wo = rho1 * w;
      acc[0] += wo * (Jx[0][0] * Jx[0][0] + Jx[1][0] * Jx[1][0] + Jx[2][0] * Jx[2][0]);
      acc[1] += wo * (Jx[0][0] * Jx[0][1] + Jx[1][0] * Jx[1][1] + Jx[2][0] * Jx[2][1]);
      acc[2] += wo * (Jx[0][0] * Jx[0][2] + Jx[1][0] * Jx[1][2] + Jx[2][0] * Jx[2][2]);
      acc[3] += wo * (Jx[0][1] * Jx[0][1] + Jx[1][1] * Jx[1][1] + Jx[2][1] * Jx[2][1]);
      acc[4] += wo * (Jx[0][1] * Jx[0][2] + Jx[1][1] * Jx[1][2] + Jx[2][1] * Jx[2][2]);
      acc[5] += wo * (Jx[0][2] * Jx[0][2] + Jx[1][2] * Jx[1][2] + Jx[2][2] * Jx[2][2]);
#pragma unroll
      for (int r = 0; r < 3; r++) acc[6 + r] -= wo * (Jx[0][r] * er[0] + Jx[1][r] * er[1] + Jx[2][r] * er[2]);
    }
  }
#pragma unroll
  for (int q = 0; q < 9; q++)
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) acc[q] += shfl_xor_d(acc[q], d);
  if (lv && sub == 0) {
    double* Ho = A.Hll + (size_t)(P.point_base + l) * 9;
    Ho[0] = acc[0]; Ho[1] = acc[1]; Ho[2] = acc[2]; Ho[3] = acc[1]; Ho[4] = acc[3]; Ho[5] = acc[4]; Ho[6] = acc[2]; Ho[7] = acc[4]; Ho[8] = acc[5];
    for (int r = 0; r < 3; r++) A.bl[(size_t)(P.point_base + l) * 3 + r] = acc[6 + r];
  }
}

// both halves of buildSystem in one launch: blocks [0, nbpose) linearise pose-major, the others point-major
__global__ __launch_bounds__(256) void ba_linearize(BaArrays A, int nbpose) {
  if (A.state[blockIdx.y].spec) return;      // a speculative twin reads its primary's linearisation (BaProb::lin_*)
  if ((int)blockIdx.x < nbpose) ba_lin_pose(A, blockIdx.x);
  else ba_lin_point(A, blockIdx.x - nbpose);
}

// chi2 total, lambda init on the first iteration of a stage (levenberg.cpp:93-97,166-180): evaluated by EVERY workgroup of ba_prep for
// itself (1 200 diagonal entries, a fixed-order reduction: the same bits everywhere); the first workgroup records it.  The phase
// stays LINEARIZE until ba_decide: the kernels of a trial run in both phases.
__device__ __forceinline__ double ba_post_lin(const BaArrays& A, const BaProb& P, BaState& S, bool record) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  double m = 0;
  if (S.iter == 0) {
    for (int a = tid; a < S.npa; a += 256) {
      const double* H = A.Hpp + (size_t)(P.lin_pose_base + A.pact[P.pose_base + a]) * 36;
      for (int j = 0; j < 6; j++) m = fmax(m, fabs(H[j * 7]));
    }
    for (int l = tid; l < P.nl; l += 256)
      if (A.lact[P.point_base + l]) {
        const double* H = A.Hll + (size_t)(P.lin_point_base + l) * 9;
        m = fmax(m, fmax(fabs(H[0]), fmax(fabs(H[4]), fabs(H[8]))));
      }
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  const double lambda = S.iter == 0 ? 1e-5 * fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) : S.lambda;
  if (record && tid == 0) {
    double chi = 0;
    for (int i = 0; i < P.np; i++) chi += A.part[P.lin_part_base + i];   // fixed order
    S.current_chi = chi;
    S.ini_chi = chi;
    if (S.iter == 0) { S.lambda = lambda; S.ni = 2; S.n_bad = 0; }
    S.trial = 0;
  }
  return lambda;
}

// The damping a problem's trial runs with.  A speculative twin (BaState::spec, see ba_decide) runs the trial its primary would run NEXT if the
// current one is rejected: g2o's retry is lambda *= ni (levenberg.cpp:139-141) on the same linearisation - this very product.
// (r06: spec = s > 1 - the s-th trial ahead: every rejection multiplies lambda by ni and doubles ni)
__device__ __forceinline__ double ba_trial_lambda(const BaState& S, double lambda, double ni) {
  for (int s = 0; s < S.spec; s++) { lambda *= ni; ni *= 2; }
  return lambda;
}

// D^-1 per point; b_s per active pose (block_solver.hpp:367-439)
__global__ __launch_bounds__(256) void ba_prep(BaArrays A) {
  const BaProb P = A.prob[blockIdx.y];
  BaState& S = A.state[blockIdx.y];
  if (S.phase != BA_PH_TRIAL && S.phase != BA_PH_LINEARIZE) return;
  const bool lin = S.phase == BA_PH_LINEARIZE;
  const double lambda0 = lin ? ba_post_lin(A, P, S, blockIdx.x == 0) : S.lambda;
  // (on the first iteration of a stage ba_post_lin's record sets ni = 2 - possibly after this workgroup has read the state)
  const double lambda = ba_trial_lambda(S, lambda0, (lin && S.iter == 0) ? 2.0 : S.ni);
  const int nbl = (P.nl + 255) / 256;
  if ((int)blockIdx.x < nbl) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= P.nl || !A.lact[P.point_base + l]) return;
    double D[9], Di[9];
    const double* H = A.Hll + (size_t)(P.lin_point_base + l) * 9;
#pragma unroll
    for (int q = 0; q < 9; q++) D[q] = H[q] + ((q % 4 == 0) ? lambda : 0.0);
    inv3(D, Di);
    double* o = A.Dinv + (size_t)(P.point_base + l) * 9;
#pragma unroll
    for (int q = 0; q < 9; q++) o[q] = Di[q];
    return;
  }
  const int a = (blockIdx.x - nbl) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (a >= S.npa) return;
  const int i = A.pact[P.pose_base + a];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int l = lane; l < P.nl; l += 64) {
    double* Wdb = A.Wd + P.W_base + ((size_t)i * P.nl + l) * 18;
    if (!A.lact[P.point_base + l]) {
      // an inactive point takes no part in the Schur complement: its block of W D^-1 is zero (ba_schur multiplies whatever is there)
#pragma unroll
      for (int q = 0; q < 9; q++) reinterpret_cast<double2*>(Wdb)[q] = make_double2(0.0, 0.0);
      continue;
    }
    double D[9], Di[9];
    const double* H = A.Hll + (size_t)(P.lin_point_base + l) * 9;
#pragma unroll
    for (int q = 0; q < 9; q++) D[q] = H[q] + ((q % 4 == 0) ? lambda : 0.0);
    inv3(D, Di);
    const double* b = A.bl + (size_t)(P.lin_point_base + l) * 3;
    const double db0 = Di[0] * b[0] + Di[1] * b[1] + Di[2] * b[2], db1 = Di[3] * b[0] + Di[4] * b[1] + Di[5] * b[2],
                 db2 = Di[6] * b[0] + Di[7] * b[1] + Di[8] * b[2];
    const double* Wb = A.W + P.lin_W_base + ((size_t)i * P.nl + l) * 18;
    double w[18];
#pragma unroll
    for (int q = 0; q < 9; q++) { const double2 v = reinterpret_cast<const double2*>(Wb)[q]; w[2 * q] = v.x; w[2 * q + 1] = v.y; }
#pragma unroll
    for (int r = 0; r < 6; r++) s[r] += w[r * 3] * db0 + w[r * 3 + 1] * db1 + w[r * 3 + 2] * db2;
    // W D^-1 of this trial for ba_schur's A side: written here, where the block and D^-1 are in registers anyway, so that ba_schur
    // takes both operands with coalesced loads (a thread per block there read 144 bytes at a 144-byte stride: 64 lines per load instruction)
    double wd[18];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
      wd[r * 3] = w0 * Di[0] + w1 * Di[3] + w2 * Di[6];
      wd[r * 3 + 1] = w0 * Di[1] + w1 * Di[4] + w2 * Di[7];
      wd[r * 3 + 2] = w0 * Di[2] + w1 * Di[5] + w2 * Di[8];
    }
#pragma unroll
    for (int q = 0; q < 9; q++) reinterpret_cast<double2*>(Wdb)[q] = make_double2(wd[2 * q], wd[2 * q + 1]);
  }
#pragma unroll
  for (int r = 0; r < 6; r++) s[r] = wave_sum(s[r]);
  if (lane == 0)
    for (int r = 0; r < 6; r++) A.bs[(size_t)P.pose_base * 6 + a * 6 + r] = A.bp[(size_t)(P.lin_pose_base + i) * 6 + r] - s[r];
}

// S(ta, tb) = [ta == tb] (H_pp + lambda I) - sum_l (W_a D_l^-1) W_b^T, lower-triangular tile pairs only.
// One workgroup = one 48 x 48 tile (8 x 8 poses); the points are streamed in chunks of 16.  Staging: thread t takes
// ONE (pose, point) pair of one side — its 6x3 W block is 18 contiguous doubles — the A side multiplies by D_l^-1 on
// the fly; then every thread accumulates a 3 x 3 register block from the two LDS tiles.
#define SCH_LC 32     // points per chunk: a chunk is one L2 round trip for the W blocks (19 chunks of 16 were 19 x 1.5 us of a 48 us kernel)
#define SCH_RS (SCH_LC * 3 + 2)   // LDS row stride in doubles: 16-byte aligned rows (128-bit operand reads), 16 rows on 64 different banks
#define SCH_LDS_BYTES (2 * 48 * SCH_RS * 8)
typedef double sch_d4 __attribute__((ext_vector_type(4)));
typedef double sch_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void ba_schur(BaArrays A, int nprob) {
  // r05: one problem per XCD at a time (workgroup b runs on XCD b % 8, every XCD has its own L2): the launch is (8 x tile pairs,
  // ceil(problems / 8)) and problem = 8 y + x % 8 - a pose tile's rows of W / W D^-1 (345 KB) are read by seven and more tile pairs
  const int prob = (int)blockIdx.y * 8 + ((int)blockIdx.x & 7);
  if (prob >= nprob) return;
  const BaProb P = A.prob[prob];
  const BaState& St = A.state[prob];
  if (St.phase != BA_PH_TRIAL && St.phase != BA_PH_LINEARIZE) return;
  const int npa = St.npa, nt = (npa + PS_BA_TILE - 1) / PS_BA_TILE;
  int ta = 0, rem = (int)blockIdx.x >> 3;   // -> (ta, tb), tb <= ta
  while (ta < nt && rem > ta) { rem -= ta + 1; ta++; }
  if (ta >= nt) return;
  const int tb = rem;
  extern __shared__ __attribute__((aligned(16))) double sch_smem[];
  double (*As)[SCH_RS] = reinterpret_cast<double (*)[SCH_RS]>(sch_smem);
  double (*Bs)[SCH_RS] = reinterpret_cast<double (*)[SCH_RS]>(sch_smem + 48 * SCH_RS);
  const int tid = threadIdx.x;
  // Staging role of this thread: side (A: W D^-1 as ba_prep left it, B: W), pose pl of the side's tile, lane lc of the pose's 16.  A
  // pose's blocks of a chunk's SCH_LC points are contiguous in memory (18 doubles per point): the 16 lanes read them as 16-byte pieces,
  // piece j * 16 + lc in step j - coalesced, and no arithmetic (until r04 a thread read its own two blocks, 144 bytes at a 144-byte
  // stride, and the A side multiplied them by D^-1 here: the load instructions alone were 2.9 of a chunk's 4.4 us on the address unit)
  const int side = tid >> 7, pl = (tid & 127) >> 4, lc = tid & 15;
  const int pc = (side == 0 ? ta : tb) * PS_BA_TILE + pl;               // compact pose index
  const bool pose_ok = pc < npa;
  const double* Wsrc = side == 0 ? A.Wd : A.W;
  const double* Wrow = Wsrc + (side == 0 ? P.W_base : P.lin_W_base) + (pose_ok ? (size_t)A.pact[P.pose_base + pc] * P.nl * 18 : 0);
  double (*dstT)[SCH_RS] = side == 0 ? As : Bs;
  // 48 x 48 tile = 3 x 3 tiles of the FP64 matrix cores (v_mfma_f64_16x16x4_f64); wave w (< 3) owns tile row w.  The vector
  // form of this contraction (3 x 3 register blocks, six LDS reads per nine FMAs) was bound by LDS instruction issue.
  const int wv = tid >> 6, ln = tid & 63, li = ln & 15, lk = ln >> 4;
  // 16-row blocks of the two tile rows that hold rows of the matrix (n = 6 npa rows in all)
  const int na = min(3, (6 * npa - ta * 48 + 15) >> 4), nc = min(3, (6 * npa - tb * 48 + 15) >> 4);
  const bool full_tile = na == 3 && nc == 3;
  sch_d4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 3; c++) acc[a][c] = sch_d4{0.0, 0.0, 0.0, 0.0};
  // the chunk's pieces are requested one chunk AHEAD, before the matrix instructions of the current chunk: the round trip to L2 runs under them
  constexpr int NPC = SCH_LC * 18 / 2 / 16;    // 16-byte pieces per lane and chunk (18)
  static_assert(SCH_LC * 18 % 32 == 0, "ba_schur: a chunk of a pose is a whole number of 16-lane steps");
  double2 w[NPC];
  auto request = [&](int l0) {
#pragma unroll
    for (int j = 0; j < NPC; j++) {
      const int e = 2 * (j * 16 + lc);          // first of the piece's two doubles inside the pose's chunk; both belong to point e / 18
      const bool ok = pose_ok && l0 + e / 18 < P.nl;
      w[j] = ok ? *reinterpret_cast<const double2*>(Wrow + (size_t)l0 * 18 + e) : make_double2(0.0, 0.0);
    }
  };
  request(0);
  for (int l0 = 0; l0 < P.nl; l0 += SCH_LC) {
    __syncthreads();   // the previous chunk's tiles are no longer being read
#pragma unroll
    for (int j = 0; j < NPC; j++) {
      const int e = 2 * (j * 16 + lc), pt = e / 18, q = e - 18 * pt;     // q even: (r, k) = (q / 3, q % 3) and its successor
      dstT[pl * 6 + q / 3][pt * 3 + q % 3] = w[j].x;
      dstT[pl * 6 + (q + 1) / 3][pt * 3 + (q + 1) % 3] = w[j].y;
    }
    __syncthreads();
    if (l0 + SCH_LC < P.nl) request(l0 + SCH_LC);
    {
      // All FOUR waves on the matrix cores: the chunk's K range is split over them (every wave all nine 16 x 16 tiles, a quarter of the
      // K-steps: 54 matrix instructions per chunk instead of 72 on three waves with the fourth idle - the kernel is bound by them,
      // PS_BA_PROFILE-style timers: 6 400 of a chunk's 9 000 cycles); the four partial tiles are added in a fixed order at the end.
      // The K index of a lane's step i is (SCH_LC * 3 / 4) lk + i on both operands (any one-to-one assignment of the chunk's K values
      // to (step, lk) is a valid contraction): a lane's operands are contiguous, two K-steps per 128-bit LDS read.
      constexpr int KL = SCH_LC * 3 / 4, KW = KL / 4;
      static_assert(4 * 2304 * 8 <= SCH_LDS_BYTES && KW % 2 == 0, "ba_schur: two K-steps per 128-bit read");
      const int k0 = KL * lk + KW * wv;
      if (full_tile) {
#pragma unroll
      for (int i = 0; i < KW; i += 2) {
        sch_d2 a2[3], b2[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { a2[a] = *reinterpret_cast<const sch_d2*>(&As[a * 16 + li][k0 + i]); b2[a] = *reinterpret_cast<const sch_d2*>(&Bs[a * 16 + li][k0 + i]); }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[a].x, b2[c].x, acc[a][c], 0, 0, 0);
            acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[a].y, b2[c].y, acc[a][c], 0, 0, 0);
          }
      }
      } else {
        // (r06) a tile row at the end of the matrix that holds fewer than 33 rows: only its first (two) 16-row blocks exist - with 49 free
        // poses (BASELINE config 4) the seventh tile row is 6 rows, and 7 of a problem's 28 workgroups ran 9 matrix tiles for one
#pragma unroll
      for (int i = 0; i < KW; i += 2) {
        sch_d2 a2[3], b2[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { a2[a] = *reinterpret_cast<const sch_d2*>(&As[a * 16 + li][k0 + i]); b2[a] = *reinterpret_cast<const sch_d2*>(&Bs[a * 16 + li][k0 + i]); }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            if (a < na && c < nc) {      // wave-uniform
              acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[a].x, b2[c].x, acc[a][c], 0, 0, 0);
              acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[a].y, b2[c].y, acc[a][c], 0, 0, 0);
            }
          }
      }
      }
    }
  }
  // the four waves' partial tiles through LDS (the staging area is free now), added in wave order
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) sch_smem[wv * 2304 + (a * 3 + c) * 256 + r * 64 + ln] = acc[a][c][r];
  __syncthreads();
  const int lda = 6 * P.np, n = 6 * npa;
  double* Sm = A.S + P.S_base;
  for (int e = tid; e < 2304; e += 256) {
    const double v = ((sch_smem[e] + sch_smem[2304 + e]) + sch_smem[2 * 2304 + e]) + sch_smem[3 * 2304 + e];
    const int t = e >> 8, r = (e >> 6) & 3, l6 = e & 63;
    const int gr = ta * 48 + (t / 3) * 16 + (l6 >> 4) + 4 * r, gc = tb * 48 + (t % 3) * 16 + (l6 & 15);   // C/D layout of the f64 MFMA: row (lane >> 4) + 4 r, column lane & 15
    if (gr >= n || gc >= n) continue;
    double h = 0;
    if (gr / 6 == gc / 6) {
      h = A.Hpp[(size_t)(P.lin_pose_base + A.pact[P.pose_base + gr / 6]) * 36 + (gr % 6) * 6 + (gc % 6)];
      if (gr == gc) h += ba_trial_lambda(St, St.lambda, St.ni);
    }
    Sm[(size_t)gr * lda + gc] = h - v;
  }
}

// Blocked unpivoted LDL^T of the reduced system S (n = 6 * active poses) and the two triangular solves, one
// workgroup per problem.  Block width NB (24 in practice): per block column
//   (1) ONE wave factors the NB x NB diagonal block with its rows in registers (lane = row, broadcasts by v_readlane,
//       no workgroup barrier), (2) every other row solves its NB-wide panel against it (thread per row, panel kept
//       in LDS), (3) the trailing matrix gets the rank-NB update from the LDS panel on the FP64 matrix cores — first the
//       tiles of the next block column, then, while wave 0 already factors the next diagonal block (1), the rest.
// Global traffic is n^3 / (3 NB) instead of n^3 / 18 with 6-wide blocks.  x_p only changes when the factorisation
// succeeds (linear_solver_eigen.h:94-120 returns false without touching x); a zero pivot = failure, like
// SimplicialLDLT.
#define SOL_T 1024
static_assert(6 * PS_BA_MAX_POSES <= SOL_T - 64, "ba_solve: one thread per row of the reduced system behind wave 0");
#ifdef PS_BA_PROFILE   // developer build: per-phase wall-clock ticks (100 MHz) of problem 0, printed by the kernel
#define SOLP_DECL long long T0 = wall_clock64(), tph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tt = T0; dgf = 0; dgl = 0
#define SOLP_MARK(k) do { const long long _n = wall_clock64(); tph[k] += _n - tt; tt = _n; } while (0)
#define SOLP_PRINT() do { if (tid == 0 && blockIdx.x == 0) printf("solve n=%d NB=%d ticks: diag + rest of trailing %lld panel (load %lld solve %lld store + forward %lld rhs %lld) trailing, next block column %lld fwd %lld bwd %lld total %lld, of which wave 0 in the diagonal blocks %lld (loads %lld, factorisation proper %lld)\n", n, NB, tph[0], tph[5], tph[6], tph[7], tph[1], tph[2], tph[3], tph[4], wall_clock64() - T0, tph[8], dgl, dgf); } while (0)
#else
#define SOLP_DECL
#define SOLP_MARK(k)
#define SOLP_PRINT()
#endif
typedef double sol_d4 __attribute__((ext_vector_type(4)));
typedef double sol_d2 __attribute__((ext_vector_type(2)));
// workgroup barrier that waits for LDS traffic only.  __syncthreads() also waits for every outstanding global access of the wave: behind
// the panel's store that is a write round trip to L2 per block step, and in the backward substitution it waited for the loads that
// had been requested ahead precisely so that they would NOT be waited for.  Used where the waves hand each other LDS data only; the
// barriers that publish trailing tiles through global memory stay __syncthreads().
__device__ __forceinline__ void sol_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NB, bool PB>
__global__ __launch_bounds__(SOL_T) void ba_solve(BaArrays A) {
  const BaProb P = A.prob[blockIdx.x];
  BaState& St = A.state[blockIdx.x];
  if (St.phase != BA_PH_TRIAL && St.phase != BA_PH_LINEARIZE) return;
  const int n = 6 * St.npa, lda = 6 * P.np, tid = threadIdx.x, lane = tid & 63;
  double* Sm = A.S + P.S_base;
  extern __shared__ __attribute__((aligned(16))) double sol_smem[];
  // the diagonal block's factor, pivots and pivot reciprocals exist twice: while the trailing update still reads block J's,
  // wave 0 already produces block J + NB's (look-ahead, see (3b))
  double* Ljj2 = sol_smem;                      // [2][NB][NB + 1]
  double* dj2 = Ljj2 + 2 * NB * (NB + 1);       // [2][NB]
  double* rdj2 = dj2 + 2 * NB;                  // [2][NB] reciprocals of the block's pivots
  double* Lt2 = rdj2 + 2 * NB;                  // [2][NB][NB] the block's multipliers again, column-major: Lt[q][c] = L[c][q]
  double* rhs = Lt2 + 2 * NB * NB;              // [n]
  double* dall = rhs + 6 * PS_BA_MAX_POSES;     // [n]
  // panel row stride: NB + 1 doubles keeps a tile's 16 rows on different banks; with PB it is NB + 2 (16-byte aligned rows: the
  // trailing update then reads a lane's NB / 4 K-steps of an operand as 128-bit words)
  constexpr int PST = PB ? NB + 2 : NB + 1;
  double* panel = dall + 6 * PS_BA_MAX_POSES;   // [rows below][PST]
  // PB: the panel a second time as -D L (= the negated un-divided values of the panel solve): the B operand of the trailing update
  // without the FP64 multiply and the read of d per K-step.  The trailing update was bound by its LDS reads (18 64-bit reads per
  // tile update, 12 600 per solve, next to 28 us of matrix-core time in a 97 us phase): with both operands K-contiguous per lane
  // (the K index of a lane's step i is (NB / 4) lk + i on BOTH operands, which is all the contraction asks for) they are 6 128-bit reads.
  double* panelB = panel + (PB ? (size_t)(n - min(NB, n) + 4) * PST : 0);
  __shared__ int fail;
#ifdef PS_BA_PROFILE
  long long dgf = 0, dgl = 0;
#endif
  if (tid == 0) fail = 0;
  // LDS keeps whatever the previous kernel on this CU left there, NaN bit patterns included, and 0 * NaN is not 0: every slot
  // that a partial last block touches with a zero multiplier (rhs beyond n, pivots beyond jb) is given a finite value first
  for (int i = tid; i < 6 * PS_BA_MAX_POSES; i += SOL_T) { rhs[i] = i < n ? A.bs[(size_t)P.pose_base * 6 + i] : 0.0; dall[i] = 1.0; }
  if (tid < 2 * NB) { dj2[tid] = 0.0; rdj2[tid] = 0.0; }
  __syncthreads();
  if (n == 0) { if (tid == 0) St.ok2 = 1; return; }
  SOLP_DECL;
  // ---- (1) diagonal block, ONE wave: lane = row, rows in registers.  Step j: every lane forms its multiplier l = a[j] / d_j;
  // the rank-1 update reads the other rows' multipliers straight out of their lanes' registers (v_readlane into a scalar pair
  // that feeds the FMA).  Entries above the diagonal are never consumed, so the update needs no masking.
  auto diag_block = [&](int J, int jb, double* Ljj, double* dj, double* rdj, double* Lt) {
#ifdef PS_BA_PROFILE
    const long long dg0b = wall_clock64();
#endif
    double a[NB];
    // (every lane loads - from a clamped position where it has no entry - and a select zeroes what does not exist: 24 loads under 24
    // different lane masks were 250 instructions of mask bookkeeping; the same in the two substitutions below)
    // (r06, measured and not kept: the next diagonal block handed over in LDS by the tiles of (3a) instead of re-read from L2 - 6.25
    // against 6.24 ms for the 8-object BA: the round trip sits under the other waves' trailing tiles)
#pragma unroll
    for (int c = 0; c < NB; c++) a[c] = Sm[(size_t)(J + min(lane, jb - 1)) * lda + min(J + c, n - 1)];      // all loads first, then the selects
#pragma unroll
    for (int c = 0; c < NB; c++) a[c] = (lane < jb && c <= lane) ? a[c] : 0.0;
#ifdef PS_BA_PROFILE
    { double sink = 0; for (int c = 0; c < NB; c++) sink += a[c]; if (sink == 1.2345e-300) fail = 2; }   // the loads have arrived
    const long long dg1 = wall_clock64();
    dgl += dg1 - dg0b;
#endif
    bool bad = false;
    // The critical path of a step is pivot -> reciprocal -> multiplier -> update of the NEXT pivot column; the other 22 updates
    // are off it.  The next pivot is therefore updated first and its reciprocal started before the rest of the row is touched.
    // No lane masks inside the loop and the reciprocals kept in a register per lane until the end (tools/ubench/ldlt_diag.hip: 11 100 ->
    // 7 600 cycles for a 24 x 24 block): every lane forms a "multiplier" a[j] / d_j, the rank-1 update uses a lane's un-divided
    // a[j] (= l d) and the multipliers of the lanes k > j only, so what rows <= j carry above the diagonal is never read.
    double myrd = 0.0, mydiag = 0.0;
    if (NB == 16) {
      double d = row_bcast_d(a[0], 0), rd = recip_d(d);
#pragma unroll
      for (int j = 0; j < NB; j++) {
        if (j < jb) {
          if (lane < 16 && d == 0) bad = true;
          myrd = lane == j ? rd : myrd;
          mydiag = lane == j ? d : mydiag;
          const double l = a[j] * rd, nl = -l;
          double dn = 1.0, rdn = 1.0;
          if (j + 1 < NB) {
            fmac_row_bcast(a[j + 1], nl, a[j], j + 1, true);
            dn = row_bcast_d(a[j + 1], j + 1);
            rdn = recip_d(dn);
          }
#pragma unroll
          for (int k = j + 2; k < NB; k++) fmac_row_bcast(a[k], nl, a[j], k, false);
          a[j] = lane > j ? l : a[j];
          d = dn; rd = rdn;
        }
      }
    } else if (NB == 24) {
      // r05: the 24 rows over TWO DPP rows - rows 0..15 on lanes 0..15, rows 16..23 on lanes 16..23.  The update of column k is one
      // v_fmac_f64_dpp row_newbcast as in the 16-row form: for k < 16 the multiplier of row k must also be reachable from DPP row 1, so
      // every step copies the lanes' (negated) multipliers 16 lanes up once (v_permlane16_swap: DPP row 1 <- DPP row 0); for k >= 16 the
      // multiplier sits in DPP row 1 itself and only rows >= 16 need it.  The pivot still travels by v_readlane (one per step).
      // tools/ubench/ldlt_diag24.hip: 4 300 against 7 600 cycles per block, same bits.  (v_permlane16_swap -> DPP read needs more than the
      // two wait states of VALU -> DPP: with s_nop 1 the lanes of DPP row 1 read the OLD register - hence the s_nop 4.)
      double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
      for (int j = 0; j < NB; j++) {
        if (j < jb) {
          if (d == 0) bad = true;
          myrd = lane == j ? rd : myrd;
          mydiag = lane == j ? d : mydiag;
          const double l = a[j] * rd, nl = -l;
          double m = nl;
          if (j < 15) {
            const int lo = __double2loint(nl), hi = __double2hiint(nl);
            const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            m = __hiloint2double(sh[0], sl[0]);
            asm volatile("s_nop 4" : "+v"(m));
          }
          double dn = 1.0, rdn = 1.0;
          if (j + 1 < NB) {
            if (j + 1 < 16) fmac_row_bcast(a[j + 1], m, a[j], j + 1, true);
            else fmac_row_bcast(a[j + 1], nl, a[j], j + 1 - 16, true);
            dn = shfl_d(a[j + 1], j + 1);
            rdn = recip_d(dn);
          }
#pragma unroll
          for (int k = j + 2; k < NB; k++) {
            if (k < 16) fmac_row_bcast(a[k], m, a[j], k, false);
            else fmac_row_bcast(a[k], nl, a[j], k - 16, false);
          }
          a[j] = lane > j ? l : a[j];
          d = dn; rd = rdn;
        }
      }
    } else {
    double d = shfl_d(a[0], 0), rd = recip_d(d);
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (j < jb) {
        if (d == 0) bad = true;
        myrd = lane == j ? rd : myrd;
        mydiag = lane == j ? d : mydiag;
        const double l = a[j] * rd;
        double dn = 1.0, rdn = 1.0;
        if (j + 1 < NB) {
          a[j + 1] -= a[j] * shfl_d(l, j + 1);
          dn = shfl_d(a[j + 1], j + 1);
          rdn = recip_d(dn);
        }
#pragma unroll
        for (int k = j + 2; k < NB; k++) a[k] -= a[j] * shfl_d(l, k);
        a[j] = lane > j ? l : a[j];
        d = dn; rd = rdn;
      }
    }
    }
#ifdef PS_BA_PROFILE
    if (myrd == 1.2345e-300) fail = 2;
    dgf += wall_clock64() - dg1;
#endif
    // The factor goes to LDS whole: every lane its row of Ljj and its column entries of Lt, above the diagonal included - what a row
    // carries there is never read (the forward substitution takes Ljj below the diagonal, the panel solve Lt[q][c] for c > q, the copy
    // to global memory c < r).  With a store per (row, column) under its own lane mask this epilogue was 650 instructions, 2.6 of the
    // 7.5 us a block took (r04 phase timers); the pivot of a lane's own row is picked up in the loop (mydiag) instead of a[lane].
    if (lane < NB) {
#pragma unroll
      for (int c = 0; c < NB; c++) { Ljj[lane * (NB + 1) + c] = a[c]; Lt[c * NB + lane] = a[c]; }
      dj[lane] = lane < jb ? mydiag : 0.0;       // (columns of a partial block that do not exist: 0)
      rdj[lane] = lane < jb ? myrd : 0.0;
      if (lane < jb) dall[J + lane] = mydiag;
    }
    if (bad && lane == 0) fail = 1;
  };
  if (tid < 64) diag_block(0, min(NB, n), Ljj2, dj2, rdj2, Lt2);
  __syncthreads();
  SOLP_MARK(0);
  for (int J = 0, cur = 0; J < n; J += NB, cur ^= 1) {
    const int jb = min(NB, n - J);
    double* Ljj = Ljj2 + cur * NB * (NB + 1);
    double* dj = dj2 + cur * NB;
    double* rdj = rdj2 + cur * NB;
    const double* Lt = Lt2 + cur * NB * NB;
    if (fail) break;
    // ---- (2) panel rows below the block: coalesced load into LDS, thread-per-row solve in LDS, coalesced store ----
    const int m0 = J + jb, m = n - m0;
    if (NB % 4 != 0) {
      for (int q = tid; q < m * NB; q += SOL_T) {      // (rows below exist under full blocks only: jb == NB, a constant divisor)
        const int i = q / NB, c = q - i * NB;
        panel[(size_t)i * PST + c] = Sm[(size_t)(m0 + i) * lda + J + c];
      }
      sol_lds_barrier();
    }
    SOLP_MARK(5);
    if (NB % 4 == 0) {
      // x L_JJ^T D = row of S: column q of the row is final once columns < q have been eliminated from it.  Right-looking order: as
      // soon as column q is final it is subtracted from every later column; every column receives its subtractions in the order
      // q = 0, 1, 2, ... whoever does them, so the result does not depend on the split.  FOUR lanes per row (a DPP quad), lane g of
      // the quad owns the columns 4 u + g: a step is the pivot column's value out of its owner's register (quad_perm broadcast) and
      // at most NB / 4 FMAs per lane, where a thread per row did up to NB - 1 - with 270 rows on 1024 threads three quarters of the
      // workgroup had no row at all, and the 24-step chain of each row was 2.7 of a block step's 16 us.  Only full blocks have rows below.
      // (r06) The rows come from global memory straight into the registers the solve works on - a quad's lanes take the row's doubles
      // 4 u + g, 32 contiguous bytes per u - and the solved row goes to the LDS panel AND back to global memory from those registers: the
      // staging pass (global -> LDS, barrier, LDS -> registers) and the store pass (LDS -> global, with a division per element to find its
      // row) are gone, one workgroup barrier and two LDS round trips per block step with them.  Same operations on the same values.
      const int g = lane & 3;
      constexpr int RP = (6 * PS_BA_MAX_POSES + SOL_T / 4 - 1) / (SOL_T / 4);       // rows a quad can own
      // (two rows of a quad requested together: 300 rows - BASELINE config 4 - are two per quad; a third set of registers spills)
      for (int r0 = 0; r0 < RP && (tid >> 2) + r0 * (SOL_T / 4) < m; r0 += 2) {
      double vin[2][NB / 4];
#pragma unroll
      for (int rp = 0; rp < 2; rp++) {
        const int i = (tid >> 2) + (r0 + rp) * (SOL_T / 4);
        const double* grow = Sm + (size_t)(m0 + min(i, m - 1)) * lda + J;
#pragma unroll
        for (int u = 0; u < NB / 4; u++) vin[rp][u] = grow[4 * u + g];
      }
#pragma unroll
      for (int rp = 0; rp < 2; rp++) {
        const int i = (tid >> 2) + (r0 + rp) * (SOL_T / 4);
        if (i >= m) break;
        double* prow = panel + (size_t)i * PST;
        double v[NB / 4];
#pragma unroll
        for (int u = 0; u < NB / 4; u++) v[u] = vin[rp][u];
#pragma unroll
        for (int uq = 0; uq < NB / 4; uq++) {
          // the multipliers of the group's four steps first (LDS), then the four dependent steps on registers
          double lt[4][NB / 4];
#pragma unroll
          for (int gq = 0; gq < 4; gq++)
#pragma unroll
            for (int u = uq; u < NB / 4; u++) lt[gq][u] = Lt[(4 * uq + gq) * NB + 4 * u + g];
#pragma unroll
          for (int gq = 0; gq < 4; gq++) {
            // x[q] * d_q, the un-divided value of column q = 4 uq + gq, from the lane that owns it
            const double xq = quad_bcast_d(v[uq], gq);
            {   // the owner's quad mates with a later column in the same group of four
              const double t = v[uq] - xq * lt[gq][uq];
              v[uq] = g > gq ? t : v[uq];
            }
#pragma unroll
            for (int u = uq + 1; u < NB / 4; u++) v[u] -= xq * lt[gq][u];
          }
        }
        // a lane's registers now hold the final un-divided values of its columns
        double* grow = Sm + (size_t)(m0 + i) * lda + J;
#pragma unroll
        for (int u = 0; u < NB / 4; u++) {
          const double lv = v[u] * rdj[4 * u + g];
          prow[4 * u + g] = lv;
          grow[4 * u + g] = lv;
          if (PB) panelB[(size_t)i * PST + 4 * u + g] = -v[u];
        }
      }
      }
    } else
    for (int i = tid; i < m; i += SOL_T) {
      // x L_JJ^T D = row of S: column q of the row is final once columns < q have been eliminated from it.  Right-looking order:
      // as soon as column q is final it is subtracted from every later column, so the 23 .. 1 updates of a step are independent
      // of each other (the dot-product order makes each column one serial FMA chain); every column still receives its
      // subtractions in the order q = 0, 1, 2, ..., i.e. the result is bit-identical.  Only full blocks have rows below them.
      // (Broadcasting the multipliers with v_readlane from registers instead of reading them from LDS was measured slower.)
      double* prow = panel + (size_t)i * PST;
      double v[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) v[c] = c < jb ? prow[c] : 0.0;
#pragma unroll
      for (int q = 0; q < NB; q++) {
        const double xq = v[q];          // x[q] * d_q, the un-divided value of column q
        // column q of L is contiguous in Lt: two multipliers per LDS read (16-byte aligned pairs start at even c)
        if (((q + 1) & 1) && q + 1 < NB) v[q + 1] -= xq * Lt[q * NB + q + 1];
#pragma unroll
        for (int c = (q + 2) & ~1; c + 1 < NB; c += 2) {
          const sol_d2 l2 = *reinterpret_cast<const sol_d2*>(&Lt[q * NB + c]);
          v[c] -= xq * l2.x;
          v[c + 1] -= xq * l2.y;
        }
        prow[q] = xq * rdj[q];
        if (PB) panelB[(size_t)i * PST + q] = -xq;
      }
    }
    sol_lds_barrier();
    SOLP_MARK(6);
    if (NB % 4 != 0)
      for (int q = tid; q < m * NB; q += SOL_T) {
        const int i = q / NB, c = q - i * NB;
        Sm[(size_t)(m0 + i) * lda + J + c] = panel[(size_t)i * PST + c];
      }
    // ... and the diagonal block's multipliers (wave 0 left them in LDS: 24 store instructions of 23 scattered rows each were 2 us of
    // its serial path per block)
    for (int q = SOL_T - 1 - tid; q < NB * NB; q += SOL_T) {      // (a constant divisor; the rows / columns a partial last block lacks are skipped)
      const int r = q / NB, c = q - r * NB;
      if (c < r && r < jb) Sm[(size_t)(J + r) * lda + J + c] = Ljj[r * (NB + 1) + c];
    }
    // forward substitution of this block column while its panel is still in LDS: y_J = L_JJ^-1 b_J, after which the rows below subtract
    // L_panel y_J - no extra pass over L in global memory.  r06: ONE wave does both, in phase (3b) below, beside the diagonal block of the
    // next step and the trailing tiles (nothing in the factorisation reads the right-hand side): the two were 1.4 us of every block step's
    // serial path - wave 0's 24-step chain, a workgroup barrier, the rows' update.  Same operations in the same order.
    auto fwd_rhs = [&]() {
      double y = lane < jb ? rhs[J + lane] : 0.0;
      double lrow[NB];   // the lane's row of the block factor, fetched before the chain of broadcasts starts
#pragma unroll
      for (int j = 0; j < NB; j++) lrow[j] = Ljj[min(lane, NB - 1) * (NB + 1) + j];
#pragma unroll
      for (int j = 0; j < NB; j++) lrow[j] = (lane > j && lane < jb) ? lrow[j] : 0.0;
#pragma unroll
      for (int j = 0; j < NB; j++) {
        if (j < jb) {
          const double yj = shfl_d(y, j);
          y -= lrow[j] * yj;                       // (lrow[j] is zero where the row has no entry)
        }
      }
      if (lane < jb) rhs[J + lane] = y;
      if (m > 0) {                                 // (only full blocks have rows below them: lane c < NB holds y_c)
        double yj[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) yj[c] = shfl_d(y, c);
        for (int i = lane; i < m; i += 64) {
          double v = rhs[m0 + i];
          const double* prow = panel + (size_t)i * PST;
          double pr[NB];
#pragma unroll
          for (int c = 0; c < NB; c++) pr[c] = prow[c];
#pragma unroll
          for (int c = 0; c < NB; c++) v -= pr[c] * yj[c];
          rhs[m0 + i] = v;
        }
      }
    };
    SOLP_MARK(7);
    SOLP_MARK(1);
    // ---- (3) trailing update S_22 -= L_21 D L_21^T on the FP64 matrix cores: 16 x 16 tiles of the lower triangle, one tile per wave
    // at a time, K = NB in steps of 4 (v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
    // C/D rows (lane >> 4) + 4 r, column lane & 15).  Both operands come from the panel in LDS (row stride NB + 1 doubles keeps
    // the 16 rows of a tile on different banks); the accumulator starts from the tile of S, so one pass reads and writes it once.
    // (3a) every wave: the tile columns that hold the NEXT block column (diagonal block + panel).  (3b) wave 0 factors the next
    // diagonal block into the other Ljj / dj / rdj buffers while the remaining waves finish the rest of the trailing matrix:
    // the serial factorisation no longer has the other fifteen waves waiting for it.
    {
      const int wave = tid >> 6, nwave = SOL_T / 64;
      const int ntile = (m + 15) >> 4;
      const int li = lane & 15, lk = lane >> 4;
      // A tile is: its 16 x 16 of S from L2 (the accumulator), the panel rows from LDS, NB / 4 matrix instructions, the tile back to L2.
      // A wave can request the accumulators of TB of its tiles before it works on the first (measured: see TB).
      auto tile_load = [&](int ti, int tj) {
        const int I0 = ti * 16, col = tj * 16 + li;
        sol_d4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = I0 + lk + 4 * r;
          acc[r] = (row < m && col <= row) ? Sm[(size_t)(m0 + row) * lda + m0 + col] : 0.0;
        }
        return acc;
      };
      auto tile_finish = [&](int ti, int tj, sol_d4 acc) {
        const int I0 = ti * 16, J0 = tj * 16;
        const int col = J0 + li;
        if (PB && NB % 8 == 0) {
          const double* pa = panel + (size_t)min(I0 + li, m - 1) * PST + (NB / 4) * lk;
          const double* pb = panelB + (size_t)min(J0 + li, m - 1) * PST + (NB / 4) * lk;
          sol_d2 a2[NB / 8], b2[NB / 8];
#pragma unroll
          for (int i = 0; i < NB / 8; i++) { a2[i] = *reinterpret_cast<const sol_d2*>(pa + 2 * i); b2[i] = *reinterpret_cast<const sol_d2*>(pb + 2 * i); }
#pragma unroll
          for (int i = 0; i < NB / 8; i++) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i].x, b2[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i].y, b2[i].y, acc, 0, 0, 0);
          }
        } else {
          const double* pa = panel + (size_t)min(I0 + li, m - 1) * PST;
          const double* pb = panel + (size_t)min(J0 + li, m - 1) * PST;
#pragma unroll
          for (int kc = 0; kc < NB / 4; kc++) {
            const int k = 4 * kc + lk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k], -dj[k] * pb[k], acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = I0 + lk + 4 * r;
          if (row < m && col <= row) Sm[(size_t)(m0 + row) * lda + m0 + col] = acc[r];
        }
      };
      // the tiles t0, t0 + stride, ... < nT of a phase, TB at a time; `decode` turns a tile number into its (row, column)
      constexpr int TB = 1;       // (4: no faster - 90 against 86 us for the two phases; the early block steps are bound by the matrix pipe, the late ones by the diagonal block)
      auto run_tiles = [&](int t0, int nT, int stride, auto decode) {
        for (int t = t0; t < nT; t += TB * stride) {
          sol_d4 acc[TB];
          int ti[TB], tj[TB];
#pragma unroll
          for (int u = 0; u < TB; u++) {
            ti[u] = 0; tj[u] = 0;
            if (t + u * stride < nT) { decode(t + u * stride, ti[u], tj[u]); acc[u] = tile_load(ti[u], tj[u]); }
          }
#pragma unroll
          for (int u = 0; u < TB; u++)
            if (t + u * stride < nT) tile_finish(ti[u], tj[u], acc[u]);
        }
      };
      constexpr int TA = (NB + 15) / 16;          // tile columns of the next block column
      const int ta = min(TA, ntile);
      int nA = 0;
      for (int c = 0; c < ta; c++) nA += ntile - c;
      run_tiles(wave, nA, nwave, [&](int t, int& ti, int& tj) {
        int c = 0, u = t;
        while (u >= ntile - c) { u -= ntile - c; c++; }
        ti = c + u; tj = c;
      });
      __syncthreads();
      SOLP_MARK(2);
      const int nt2 = ntile - ta, nB = nt2 > 0 ? nt2 * (nt2 + 1) / 2 : 0;
      if (wave == 0) {
#ifdef PS_BA_PROFILE
        const long long dg0 = wall_clock64();
#endif
        if (m > 0) diag_block(m0, min(NB, m), Ljj2 + (cur ^ 1) * NB * (NB + 1), dj2 + (cur ^ 1) * NB, rdj2 + (cur ^ 1) * NB, Lt2 + (cur ^ 1) * NB * NB);
#ifdef PS_BA_PROFILE
        tph[8] += wall_clock64() - dg0;
#endif
      } else {
        if (wave == nwave - 1) fwd_rhs();
        run_tiles(wave - 1, nB, nwave - 1, [&](int t, int& ti, int& tj) {
          int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
          while (i * (i + 1) / 2 > t) i--;
          while ((i + 1) * (i + 2) / 2 <= t) i++;
          ti = i + ta; tj = t - i * (i + 1) / 2 + ta;
        });
      }
    }
    __syncthreads();
    SOLP_MARK(0);
  }
  if (fail) { if (tid == 0) St.ok2 = 0; return; }
  for (int i = tid; i < n; i += SOL_T) rhs[i] /= dall[i];
  __syncthreads();
  SOLP_MARK(3);
  // ---- backward substitution L^T x = z ----
  // Per block column, from the last one: wave 0 solves the block's unit upper triangle (lane = column r: x_r -= sum_{c > r}
  // L[J+c][J+r] x_c), then the rows above subtract L_panel^T x_J.  The factor lives in global memory (L2); both parties request
  // their part of it BEFORE they have to wait for the other one - waves 1.. fetch their panel column while wave 0 solves, wave 0
  // fetches the next block while they update - so a step costs one round trip instead of two.
  {
    const int Jlast = ((n - 1) / NB) * NB;
    double Lc[NB];
    if (tid < 64) {
      const int jb = min(NB, n - Jlast);
#pragma unroll
      for (int c = 0; c < NB; c++) Lc[c] = Sm[(size_t)min(Jlast + c, n - 1) * lda + min(Jlast + lane, n - 1)];
#pragma unroll
      for (int c = 0; c < NB; c++) Lc[c] = (c < jb && lane < c) ? Lc[c] : 0.0;
    }
    for (int J = Jlast; J >= 0; J -= NB) {
      const int jb = min(NB, n - J);
      double lc[NB];
      const int k = tid - 64;   // row of the update: J <= 6 * PS_BA_MAX_POSES - NB rows fit the 960 threads behind wave 0
      if (tid < 64) {
        double x = lane < jb ? rhs[J + lane] : 0.0;
#pragma unroll
        for (int c = NB - 1; c >= 1; c--) {
          if (c < jb) {
            const double xc = shfl_d(x, c);
            x -= Lc[c] * xc;                       // (Lc[c] is zero for the lanes >= c)
          }
        }
        if (lane < jb) rhs[J + lane] = x;
        if (J >= NB) {   // the block above is a full one
#pragma unroll
          for (int c = 0; c < NB; c++) Lc[c] = Sm[(size_t)(J - NB + c) * lda + J - NB + min(lane, NB - 1)];
#pragma unroll
          for (int c = 0; c < NB; c++) Lc[c] = lane < c ? Lc[c] : 0.0;
        }
      } else if (k < J) {
#pragma unroll
        for (int c = 0; c < NB; c++) lc[c] = c < jb ? Sm[(size_t)(J + c) * lda + k] : 0.0;
      }
      sol_lds_barrier();
      if (tid >= 64 && k < J) {
        double v = rhs[k];
#pragma unroll
        for (int c = 0; c < NB; c++) if (c < jb) v -= lc[c] * rhs[J + c];
        rhs[k] = v;
      }
      sol_lds_barrier();
    }
  }
  SOLP_MARK(4);
  for (int i = tid; i < n; i += SOL_T) A.xp[(size_t)P.pose_base * 6 + i] = rhs[i];
  if (tid == 0) St.ok2 = 1;
  SOLP_PRINT();
}

// x_l, push (backup) + oplus on every active vertex, partial sums of x . (lambda x + b)
__global__ __launch_bounds__(256) void ba_update(BaArrays A) {
  const BaProb P = A.prob[blockIdx.y];
  const BaState& S = A.state[blockIdx.y];
  if (S.phase != BA_PH_TRIAL && S.phase != BA_PH_LINEARIZE) return;
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int nbl = (P.nl + PS_BA_UPD_PPB - 1) / PS_BA_UPD_PPB;
  const double lambda = ba_trial_lambda(S, S.lambda, S.ni);
  double sc = 0;
  if ((int)blockIdx.x < nbl) {
    // 16 lanes per point: each takes every 16th free pose of c = b_l - sum_a W_a^T x_a, then the group adds up (the thread-per-
    // point form walked all the poses with one dependent L2 round trip each)
    const int l = blockIdx.x * PS_BA_UPD_PPB + (tid >> 4), sub = tid & 15;
    const bool lv = l < P.nl;
    const bool act = lv && A.lact[P.point_base + l];
    double c[3] = {0, 0, 0};
    if (act && S.ok2) {
      for (int a = sub; a < S.npa; a += 16) {
        const double* Wb = A.W + P.lin_W_base + ((size_t)A.pact[P.pose_base + a] * P.nl + l) * 18;
        const double* xp = A.xp + (size_t)P.pose_base * 6 + a * 6;
#pragma unroll
        for (int r = 0; r < 6; r++) { c[0] -= Wb[r * 3] * xp[r]; c[1] -= Wb[r * 3 + 1] * xp[r]; c[2] -= Wb[r * 3 + 2] * xp[r]; }
      }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) c[q] += shfl_xor_d(c[q], d);
    if (lv && sub == 0) {
      double* X = A.points + (size_t)(P.point_base + l) * 3;
      double* Xb = A.points_bak + (size_t)(P.point_base + l) * 3;
      Xb[0] = X[0]; Xb[1] = X[1]; Xb[2] = X[2];
      if (act) {
        double* xl = A.xl + (size_t)(P.point_base + l) * 3;
        const double* b = A.bl + (size_t)(P.lin_point_base + l) * 3;
        if (S.ok2) {
          c[0] += b[0]; c[1] += b[1]; c[2] += b[2];
          const double* Di = A.Dinv + (size_t)(P.point_base + l) * 9;
          xl[0] = Di[0] * c[0] + Di[1] * c[1] + Di[2] * c[2];
          xl[1] = Di[3] * c[0] + Di[4] * c[1] + Di[5] * c[2];
          xl[2] = Di[6] * c[0] + Di[7] * c[1] + Di[8] * c[2];
        }
        X[0] += xl[0]; X[1] += xl[1]; X[2] += xl[2];
        sc = xl[0] * (lambda * xl[0] + b[0]) + xl[1] * (lambda * xl[1] + b[1]) + xl[2] * (lambda * xl[2] + b[2]);
      }
    }
  } else {
    const int i = (blockIdx.x - nbl) * 256 + tid;
    if (i < P.np) {
      double* pp = A.poses + (size_t)(P.pose_base + i) * 7;
      double* pb = A.poses_bak + (size_t)(P.pose_base + i) * 7;
      for (int q = 0; q < 7; q++) pb[q] = pp[q];
      const int a = A.pidx[P.pose_base + i];
      if (a >= 0) {
        const double* xp = A.xp + (size_t)P.pose_base * 6 + a * 6;
        const double* b = A.bp + (size_t)(P.lin_pose_base + i) * 6;
        double u[6] = {xp[0], xp[1], xp[2], xp[3], xp[4], xp[5]};
        for (int r = 0; r < 6; r++) sc += u[r] * (lambda * u[r] + b[r]);
        const bool nrp = (A.pose_flags[P.pose_base + i] >> 1) & 1;
        if (nrp) { u[0] = 0; u[1] = 0; }   // VertexSE3Fix::oplusImpl, whether_fixrollpitch (g2o_Object.cc:190-213)
        store_pose(pp, se3_mul(se3_exp(u, nrp), load_pose(pp)));
      }
    }
  }
  sc = wave_sum(sc);
  if ((tid & 63) == 0) red[tid >> 6] = sc;
  __syncthreads();
  if (tid == 0) A.part[P.part_base + P.np + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// computeActiveErrors + activeRobustChi2 at the trial estimate
__global__ __launch_bounds__(256) void ba_error_k(BaArrays A, int err_part_off) {
  const BaProb P = A.prob[blockIdx.y];
  const BaState& S = A.state[blockIdx.y];
  if (S.phase != BA_PH_TRIAL && S.phase != BA_PH_LINEARIZE) return;
  __shared__ double red[4];
  const int tid = threadIdx.x, e = blockIdx.x * 256 + tid;
  double chi = 0;
  if (e < P.ne) {
    const int ge = P.edge_base + e;
    const uint8_t st = A.e_state[ge];
    if (!(st & ES_LVL1)) {
      const bool mono = st & ES_MONO;
      const Se3 T = load_pose(A.poses + (size_t)(P.pose_base + A.e_pose[ge]) * 7);
      double p[3], er[3];
      ba_error(T, P, A.points + (size_t)(P.point_base + A.e_point[ge]) * 3, A.e_obs + (size_t)ge * 3, mono, p, er);
      const double chi2 = ba_edge_chi2(er, (double)A.e_is2[ge]);
      A.chi2c[ge] = chi2;
      double rho0 = chi2, rho1;
      if (S.robust) huber(chi2, mono ? DELTA_MONO : DELTA_STEREO, rho0, rho1);
      chi = rho0;
    }
  }
  chi = wave_sum(chi);
  if ((tid & 63) == 0) red[tid >> 6] = chi;
  __syncthreads();
  if (tid == 0) A.part[P.part_base + err_part_off + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// gain ratio, lambda schedule, accept / pop, iteration and stage control (levenberg.cpp:121-161)
// `twins`: the batch holds every problem twice, primary 2 b and speculative twin 2 b + 1, and this workgroup decides for the pair.
// Both start a trial round from the same estimate and the same linearisation (each computes its own copy with the same code on the
// same data); the primary runs the trial g2o would run now, the twin the one g2o would run next if this one is rejected
// (lambda *= ni: the retry sequence is fixed before the first trial ends).  In BASELINE config 4 every second iteration is
// "first trial rejected, second accepted" (profiles/r05_trial_histogram.txt): the pair finishes it in one round of kernels instead of
// two, for the price of a batch twice as large (+17 % per round at 8 objects).  The decision below replays the two trials in
// sequence order with g2o's bookkeeping (trial counts, lambda / ni, the stop rules), then makes the twins identical again: the
// estimate the sequence ends with, the increments and - when a stage ends here - the cached chi2 of the LAST trial g2o ran.
// A twin whose factorisation failed is ignored (its "whatever x holds" update would need the primary's x of this same round): the
// primary runs that lambda itself in the next round.
// r06: `grp` members per problem instead of a pair (2, 3 or 4: member s runs the s-th trial ahead) - BASELINE config 4 has 1 / 2 / 3 / 4 trials in
// 29 / 62 / 22 / 7 of its 120 iterations; with three members the 22 three-trial iterations take one round of kernels as well.
__global__ __launch_bounds__(256) void ba_decide(BaArrays A, int err_part_off, int grp) {
  const int pa = grp * blockIdx.x;
  const bool twins = grp > 1;
  const BaProb P = A.prob[pa];
  BaState& S = A.state[pa];
  if (S.phase != BA_PH_TRIAL && S.phase != BA_PH_LINEARIZE) return;
  __shared__ int s_acc, s_last;
  __shared__ double s_part[4][1024];
  const int tid = threadIdx.x;
  // the partial sums of the trial (scale: one per block of ba_update, chi2: one per block of ba_error_k) come in with one round trip for
  // all of them; thread 0 then adds them in the order it always did (it used to fetch them one dependent load after the other: 15 us)
  const int nbl = (P.nl + PS_BA_UPD_PPB - 1) / PS_BA_UPD_PPB, nbp = (P.np + 255) / 256, nbe = (P.ne + 255) / 256;
  const int ns = nbl + nbp, staged = ns + nbe <= 1024;
  if (staged)
    for (int b = tid; b < ns + nbe; b += 256)
      for (int w = 0; w < grp; w++) {
        const int pbw = A.prob[pa + w].part_base;
        s_part[w][b] = b < ns ? A.part[pbw + P.np + b] : A.part[pbw + err_part_off + (b - ns)];
      }
  __syncthreads();
  if (tid == 0) {
    int accepted = -1, last = 0;
    double rho = 0;
    const int act = min(grp, S.depth + 1);               // members that ran a trial this round
    for (int w = 0; w < act; w++) {
      const BaState& T = A.state[pa + w];
      if (w >= 1 && !T.ok2) break;                       // a speculative member's trial is void: the primary repeats it
      const int pbase = A.prob[pa + w].part_base;
      double scale = 0, temp = 0;
      if (staged) {
        for (int b = 0; b < ns; b++) scale += s_part[w][b];
        for (int b = 0; b < nbe; b++) temp += s_part[w][ns + b];
      } else {
        for (int b = 0; b < ns; b++) scale += A.part[pbase + P.np + b];
        for (int b = 0; b < nbe; b++) temp += A.part[pbase + err_part_off + b];
      }
      if (!T.ok2) temp = DBL_MAX;
      rho = (S.current_chi - temp) / (scale + 1e-3);
      last = w;
      if (rho > 0 && isfinite(temp)) {
        double alpha = 1. - se3_cube(2 * rho - 1);
        alpha = fmin(alpha, 2. / 3.);
        S.lambda *= fmax(1. / 3., alpha);
        S.ni = 2;
        S.current_chi = temp;
        accepted = w;
      } else {
        S.lambda *= S.ni;
        S.ni *= 2;
      }
      S.trial++;
      S.trials_done++;
      if (!(rho < 0 && S.trial < 10)) break;             // the iteration is over
    }
    S.rho = rho;
    s_acc = accepted; s_last = last;
    if (!(rho < 0 && S.trial < 10)) {   // the iteration is over
      if (S.ntrace < PS_BA_TRACE) {
        double* tr = A.trace + ((size_t)pa * PS_BA_TRACE + S.ntrace) * 3;
        tr[0] = S.current_chi; tr[1] = S.lambda; tr[2] = S.trial;
      }
      S.ntrace++;
      S.iters_done++;
      bool terminate = (S.trial == 10 || rho == 0);
      if (!terminate) {
        if ((S.ini_chi - S.current_chi) * 1e3 < S.ini_chi) S.n_bad++; else S.n_bad = 0;
        if (S.n_bad >= 3) terminate = true;
      }
      S.iter++;
      if (terminate || S.iter >= S.max_iter) { S.stage++; S.phase = BA_PH_BEGIN; }
      else S.phase = BA_PH_LINEARIZE;
    } else S.phase = BA_PH_TRIAL;   // another damping trial on the same linearisation
    // (r06, measured and not kept: an ADAPTIVE number of members per round - as many as the last iteration needed trials, all of them after a
    // round of rejections, the rest parked in BA_PH_IDLE.  SURVEY's config 4 at 8 objects: 18 rounds where the fixed twin takes 20 and four
    // fixed members 15, at nearly the four members' cost per round: 6.35 against 5.66 / 5.10 ms.  Trial counts do not come in runs.)
    S.depth = grp - 1;
  }
  __syncthreads();
  const int acc = s_acc, last = s_last;
  if (!twins) {
    if (acc < 0) {   // _optimizer->pop()
      for (int q = tid; q < P.np * 7; q += 256) A.poses[(size_t)P.pose_base * 7 + q] = A.poses_bak[(size_t)P.pose_base * 7 + q];
      for (int q = tid; q < P.nl * 3; q += 256) A.points[(size_t)P.point_base * 3 + q] = A.points_bak[(size_t)P.point_base * 3 + q];
    }
  } else {
    // the estimate the sequence ends with: the accepted trial's, or the one before the trials (pop) - into every member
    const double* ps = acc >= 0 ? A.poses + (size_t)A.prob[pa + acc].pose_base * 7 : A.poses_bak + (size_t)P.pose_base * 7;
    const double* xs = acc >= 0 ? A.points + (size_t)A.prob[pa + acc].point_base * 3 : A.points_bak + (size_t)P.point_base * 3;
    // g2o's increment vectors persist between trials (a failed factorisation updates with what they hold): those of the last trial that ran;
    // the edge errors g2o has cached are those of the last trial it ran too: they are read when a stage ends (classification / erase list) -
    // the next linearisation overwrites them otherwise
    const BaProb F = A.prob[pa + last];
    for (int q = tid; q < P.np * 7; q += 256) {
      const double v = ps[q];
      for (int w = 0; w < grp; w++) if (w != acc) A.poses[(size_t)A.prob[pa + w].pose_base * 7 + q] = v;
    }
    for (int q = tid; q < P.nl * 3; q += 256) {
      const double v = xs[q];
      for (int w = 0; w < grp; w++) if (w != acc) A.points[(size_t)A.prob[pa + w].point_base * 3 + q] = v;
    }
    for (int q = tid; q < P.np * 6; q += 256) {
      const double v = A.xp[(size_t)F.pose_base * 6 + q];
      for (int w = 0; w < grp; w++) if (w != last) A.xp[(size_t)A.prob[pa + w].pose_base * 6 + q] = v;
    }
    for (int q = tid; q < P.nl * 3; q += 256) {
      const double v = A.xl[(size_t)F.point_base * 3 + q];
      for (int w = 0; w < grp; w++) if (w != last) A.xl[(size_t)A.prob[pa + w].point_base * 3 + q] = v;
    }
    if (S.phase == BA_PH_BEGIN)
      for (int e = tid; e < P.ne; e += 256) {
        const double v = A.chi2c[F.edge_base + e];
        for (int w = 0; w < grp; w++) if (w != last) A.chi2c[A.prob[pa + w].edge_base + e] = v;
      }
    __syncthreads();
    if (tid == 0)    // the speculative members continue from the primary's state; those beyond the round's depth sit it out
      for (int w = 1; w < grp; w++) {
        BaState& T = A.state[pa + w];
        const int ok2 = T.ok2;
        T = S;
        T.spec = w; T.ok2 = ok2;
      }
  }
  // a stage ended: its successor's entry (classification, active sets) right here instead of in a launch of its own per global step
  __syncthreads();
  if (S.phase == BA_PH_BEGIN) {
    ba_stage_entry(A, pa);
    if (twins) {
      // the other members start the stage from the same estimate and the same cached errors: they take over what the entry decided (edge
      // levels / erase list, active sets, cleared increments) instead of working it out again (50 us of classification and compaction)
      __syncthreads();
      for (int w = 1; w < grp; w++) {
        const BaProb Q = A.prob[pa + w];
        for (int e = tid; e < P.ne; e += 256) { A.e_state[Q.edge_base + e] = A.e_state[P.edge_base + e]; A.erase[Q.edge_base + e] = A.erase[P.edge_base + e]; }
        for (int i = tid; i < P.np; i += 256) { A.pidx[Q.pose_base + i] = A.pidx[P.pose_base + i]; A.pact[Q.pose_base + i] = A.pact[P.pose_base + i]; }
        for (int l = tid; l < P.nl; l += 256) A.lact[Q.point_base + l] = A.lact[P.point_base + l];
        for (int q = tid; q < P.np * 6; q += 256) A.xp[(size_t)Q.pose_base * 6 + q] = A.xp[(size_t)P.pose_base * 6 + q];
        for (int q = tid; q < P.nl * 3; q += 256) A.xl[(size_t)Q.point_base * 3 + q] = A.xl[(size_t)P.point_base * 3 + q];
      }
      if (tid == 0)
        for (int w = 1; w < grp; w++) {
          BaState& T = A.state[pa + w];
          T = S;
          T.spec = w;
          if (S.phase == BA_PH_DONE) atomicAdd(A.ndone, 1);
        }
    }
  }
}

}  // namespace

// one "global step": every unfinished problem advances by one LM trial (plus linearisation / stage entry
// when it is due).  max_* are maxima over the batch.
extern "C" void psk_ba_global_step(const BaArrays* A, int nprob, int max_np, int max_nl, int max_ne, int max_tilepairs,
                                   int max_free, int first, int grp, hipStream_t st) {
  const int nbl = (max_nl + 255) / 256, nbp = (max_np + 255) / 256, nbe = (max_ne + 255) / 256;
  const int nblu = (max_nl + PS_BA_UPD_PPB - 1) / PS_BA_UPD_PPB;   // ba_update's point blocks
  const int err_off = max_np + nblu + nbp;   // layout of `part`: [np chi partials][update partials][error partials]
  if (first) hipLaunchKernelGGL(ba_begin, dim3(nprob), dim3(256), 0, st, *A);
  hipLaunchKernelGGL(ba_linearize, dim3((max_np + 3) / 4 + (max_nl + 15) / 16, nprob), dim3(256), 0, st, *A, (max_np + 3) / 4);
  hipLaunchKernelGGL(ba_prep, dim3(nbl + (max_np + 3) / 4, nprob), dim3(256), 0, st, *A);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_schur), hipFuncAttributeMaxDynamicSharedMemorySize, SCH_LDS_BYTES);
  hipLaunchKernelGGL(ba_schur, dim3(8 * max_tilepairs, (nprob + 7) / 8), dim3(256), SCH_LDS_BYTES, st, *A, nprob);
  {
    const int n_max = 6 * max_free;
    auto lds = [&](int nb, bool pb) { return (size_t)(2 * nb * (nb + 1) + 4 * nb + 2 * nb * nb + 12 * PS_BA_MAX_POSES + (size_t)(pb ? 2 * (nb + 2) : nb + 1) * (n_max > nb ? n_max - nb + 4 : 4) + 8) * sizeof(double); };
    // > 64 KB of dynamic LDS has to be requested per kernel
    static const int force_nb = getenv("PS_BA_NB") ? atoi(getenv("PS_BA_NB")) : 0;
    static const bool no_pb = getenv("PS_BA_NO_PB") != nullptr;   // developer knob: one panel, the B operand scaled on the fly
    if (force_nb == 16 && lds(16, true) <= 156 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_solve<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(16, true));
      hipLaunchKernelGGL((ba_solve<16, true>), dim3(nprob), dim3(SOL_T), lds(16, true), st, *A);
    } else if (force_nb == 48 && lds(48, false) <= 150 * 1024) {   // measured slower than 24 at n = 294 (single-wave diagonal factor)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_solve<48, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(48, false));
      hipLaunchKernelGGL((ba_solve<48, false>), dim3(nprob), dim3(SOL_T), lds(48, false), st, *A);
    } else if (force_nb != 12 && !no_pb && lds(24, true) <= 156 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_solve<24, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(24, true));
      hipLaunchKernelGGL((ba_solve<24, true>), dim3(nprob), dim3(SOL_T), lds(24, true), st, *A);
    } else if (force_nb != 12 && lds(24, false) <= 150 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_solve<24, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(24, false));
      hipLaunchKernelGGL((ba_solve<24, false>), dim3(nprob), dim3(SOL_T), lds(24, false), st, *A);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_solve<12, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(12, false));
      hipLaunchKernelGGL((ba_solve<12, false>), dim3(nprob), dim3(SOL_T), lds(12, false), st, *A);
    }
  }
  hipLaunchKernelGGL(ba_update, dim3(nblu + nbp, nprob), dim3(256), 0, st, *A);
  hipLaunchKernelGGL(ba_error_k, dim3(nbe, nprob), dim3(256), 0, st, *A, err_off);
  hipLaunchKernelGGL(ba_decide, dim3(nprob / grp), dim3(256), 0, st, *A, err_off, grp);
}
// the stage-2 pass of ba_begin (erase list) needs one more launch once every problem left its last trial
extern "C" void psk_ba_finalize(const BaArrays* A, int nprob, hipStream_t st) {
  hipLaunchKernelGGL(ba_begin, dim3(nprob), dim3(256), 0, st, *A);
}
