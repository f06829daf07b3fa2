// Glue kernels of the device-resident lockstep tracker: the host side of Tracking::Track between the hot-path calls
// (pointslot_amd/host/StereoOdometry.h: OdoSequence, which documents the mapping to /root/reference/src/Tracking.cc:2840-3160,
// 1260-1286 and src/Frame.cc:1636-1656,1686-1743,2505-2519) moved onto the device, so that a frame's chain
//   ExtractORB x 2 -> ComputeStereoMatches -> SearchByProjection(cur, last) -> PoseOptimization -> SearchLocalPoints ->
//   SearchByProjection(F, points) -> PoseOptimization
// is one stream of launches with no host round trip and no PCIe traffic besides the images.  One 256-thread workgroup per
// sequence and step of the state machine; a sequence that is not in the phase a kernel serves leaves it immediately, and the
// matcher / optimiser problems of such a sequence are empty (nq = nt = 0, no edges).
// float arithmetic is written operation by operation exactly as the host class evaluates it (this file is compiled with
// -ffp-contract=off): the trajectories of the two drivers are compared bit for bit in tests/test_track_device_gpu.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pointslot_hip.h"
#include "track_plan.h"
#include "se3.h"

namespace {

#ifndef TRK_T
#define TRK_T 256       // threads per sequence.  The glue kernels are chains of short dependent phases and more lanes shorten each (alone: 0.33 ->
                        // 0.23 ms per step with 1024), but a 1024-thread workgroup needs sixteen free wave slots on ONE CU at the same moment: beside
                        // the other lockstep groups' kernels trk_finish waited 500 us for its turn (89 us alone).  r04: 256 -> +1.3 % frames/s
#endif
#ifndef TRK_T_SMALL
#define TRK_T_SMALL 1024   // ... and a handle of a few sequences (one per GPU: BASELINE configs[4]) has the chip to itself: every kernel is instantiated for
#endif                     // both widths and the launchers below pick by the handle's size.  Results do not depend on the width (integer sums, ordered scans, minima)
#ifndef TRK_SMALL_S
#define TRK_SMALL_S 32
#endif

__device__ __forceinline__ void mul4(const float* a, const float* b, float* o) {   // OdoSequence::mul4
  float t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      float acc = 0;
      for (int k = 0; k < 4; k++) acc += a[4 * r + k] * b[4 * k + c];
      t[4 * r + c] = acc;
    }
  for (int i = 0; i < 16; i++) o[i] = t[i];
}

// Frame::UnprojectStereo (Frame.cc:2505-2519) as OdoSequence::unproject evaluates it
__device__ __forceinline__ void unproject(const TrkCam& C, const float* T, float x, float y, float z, float* X) {
  const float xc = (x - C.cx) * z * C.inv_fx, yc = (y - C.cy) * z * C.inv_fy;
  float Ow[3];
  for (int r = 0; r < 3; r++) Ow[r] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
  for (int r = 0; r < 3; r++) X[r] = (T[r] * xc + T[4 + r] * yc + T[8 + r] * z) + Ow[r];
}

// sum of v over the workgroup (every thread gets it); red: NTH / 64 ints of LDS
template <int NTH>
__device__ __forceinline__ int block_sum_i(int v, int* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < NTH / 64; w++) t += red[w];
  return t;
}

// exclusive prefix of v over the workgroup in thread order; *total = the sum
template <int NTH>
__device__ __forceinline__ int block_scan_excl(int v, int* red, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  __syncthreads();
  if (lane == 63) red[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += red[w];
  int t = 0;
#pragma unroll
  for (int w = 0; w < NTH / 64; w++) t += red[w];
  *total = t;
  return base + incl - v;
}

__device__ __forceinline__ void set_identity(float* m) {
  for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

__device__ void fill_search_common(PjProb& d, const TrkArrays& A, int s, int nt, int nq) {
  d.t_off = s * A.cap; d.nt = nt; d.q_off = s * A.cap; d.c_off = d.q_off; d.nq = nq; d.grid_off = s * (PS_TRK_NCELL + 1);
  d.min_x = 0.f; d.min_y = 0.f; d.gw_inv = A.cam.gw_inv; d.gh_inv = A.cam.gh_inv;
  d.fx = A.cam.fx; d.fy = A.cam.fy; d.cx = A.cam.cx; d.cy = A.cam.cy; d.mbf = A.cam.mbf; d.mb = A.cam.mb;
  d.bounds[0] = 0.f; d.bounds[1] = (float)A.cam.w; d.bounds[2] = 0.f; d.bounds[3] = (float)A.cam.h;
  for (int l = 0; l < 8; l++) d.scale[l] = l < A.cam.nlevels ? A.cam.sf[l] : 1.f;
  d.mono = 0; d.use_bbox = 0;
}

__device__ void empty_problem(PjProb* p, const TrkArrays& A, int s) {
  PjProb d;
  for (int i = 0; i < (int)(sizeof(PjProb) / 4); i++) ((int32_t*)&d)[i] = 0;
  fill_search_common(d, A, s, 0, 0);
  *p = d;
}

// OdoSequence::fail
__device__ void seq_fail(const TrkArrays& A, int s, int step, TrkStat& st) {
  TrkSeq& q = A.seq[s];
  q.state = TRK_LOST; q.have_velocity = 0; q.phase = TRK_PH_IDLE;
  // the chain's remaining pose_lm launches must see no edges for this sequence (the host loop stops calling the optimiser here)
  A.po_vert[s] = PoVertex{(int32_t)(s * A.cap), (int32_t)(s * A.cap)};
  st.state = TRK_LOST; st.tracked = 0;
  A.stats[(size_t)step * A.S + s] = st;
  float* tr = A.traj + ((size_t)step * A.S + s) * 16;
  for (int i = 0; i < 16; i++) tr[i] = 0.f;
}

// copies sequence s of frame `src` over frame `dst` (last = F)
template <int NTH>
__device__ void copy_frame(const TrkArrays& A, const TrkFrame& dst, const TrkFrame& src, int s) {
  const int tid = threadIdx.x, n = src.n[s];
  const size_t b = (size_t)s * A.cap;
  for (int i = tid; i < n; i += NTH) {
    dst.x[b + i] = src.x[b + i]; dst.y[b + i] = src.y[b + i]; dst.angle[b + i] = src.angle[b + i];
    dst.uright[b + i] = src.uright[b + i]; dst.depth[b + i] = src.depth[b + i]; dst.octave[b + i] = src.octave[b + i];
    dst.mp_id[b + i] = src.mp_id[b + i]; dst.mp_valid[b + i] = src.mp_valid[b + i]; dst.mp_observed[b + i] = src.mp_observed[b + i];
    dst.outlier[b + i] = src.outlier[b + i];
    for (int c = 0; c < 3; c++) dst.xw[3 * (b + i) + c] = src.xw[3 * (b + i) + c];
  }
  const uint4* sd = reinterpret_cast<const uint4*>(src.desc + b * 32);
  uint4* dd = reinterpret_cast<uint4*>(dst.desc + b * 32);
  for (int i = tid; i < 2 * n; i += NTH) dd[i] = sd[i];
  // (the grid of the last frame is never read: SearchByProjection walks the CURRENT frame's grid)
  if (tid < 16) dst.tcw[s * 16 + tid] = src.tcw[s * 16 + tid];
  if (tid == 0) dst.n[s] = n;
}

// ---------------------------------------------------------------------------------------------------------------------
// Step 1: Frame::Frame after the extractors (per-keypoint arrays, AssignFeaturesToGrid, Frame.cc:1636-1656,2027-2037), then
// StereoInitialization (Tracking.cc:2840-2910) or UpdateLastFrame + the motion-model prediction + the first
// SearchByProjection(cur, last, th = 7) problem (Tracking.cc:2971-3048).
// ---------------------------------------------------------------------------------------------------------------------
template <int NTH>
__global__ __launch_bounds__(NTH) void trk_begin(TrkArrays A, int step) {
  __shared__ int cnt[PS_TRK_NCELL + 1];
  __shared__ int cursor[PS_TRK_NCELL];
  __shared__ float sdepth[4352];
  __shared__ int red[NTH / 64];
  __shared__ unsigned long long far_red[NTH / 64];
  __shared__ float pose_pred[16];
  const int s = blockIdx.x, tid = threadIdx.x;
  const TrkCam& C = A.cam;
  TrkSeq& q = A.seq[s];
  const size_t b = (size_t)s * A.cap;
  const ps_keypoint* kps = (const ps_keypoint*)A.orb_kps + (size_t)(2 * s) * A.kp_cap;
  const uint8_t* desc = A.orb_desc + (size_t)(2 * s) * A.kp_cap * 32;
  int N = A.orb_counts[2 * s];
  N = N < A.cap ? N : A.cap;
  TrkStat st;
  for (int i = 0; i < (int)(sizeof(TrkStat) / 4); i++) ((int32_t*)&st)[i] = 0;
  st.n = N;

  // ---- per-keypoint arrays ----
  for (int i = tid; i <= PS_TRK_NCELL; i += NTH) cnt[i] = 0;
  __syncthreads();
  {
    // with an instance mask (SLOT.MODE 4) Frame::AssignFeatures keeps the keypoints on background pixels, in order (Frame.cc:811-822)
    const uint8_t* M = A.idmask ? A.idmask + (size_t)s * A.mask_pitch : nullptr;
    int kept = 0;
    for (int i0 = 0; i0 < N; i0 += NTH) {
      const int i = i0 + tid;
      ps_keypoint k;
      bool keep = i < N;
      if (keep) {
        k = kps[i];
        if (M) keep = M[(size_t)(int)k.y * A.mask_stride + (int)k.x] == 0;
      }
      int pos = i;
      if (M) {
        int total;
        pos = kept + block_scan_excl<NTH>(keep ? 1 : 0, red, &total);
        kept += total;
      }
      if (keep) {
        A.cur.x[b + pos] = k.x; A.cur.y[b + pos] = k.y; A.cur.angle[b + pos] = k.angle; A.cur.octave[b + pos] = k.octave;
        A.cur.uright[b + pos] = A.orb_uright[(size_t)s * A.kp_cap + i]; A.cur.depth[b + pos] = A.orb_depth[(size_t)s * A.kp_cap + i];
        A.cur.xw[3 * (b + pos)] = 0.f; A.cur.xw[3 * (b + pos) + 1] = 0.f; A.cur.xw[3 * (b + pos) + 2] = 0.f;
        A.cur.mp_valid[b + pos] = 0; A.cur.mp_observed[b + pos] = 0; A.cur.outlier[b + pos] = 0; A.cur.mp_id[b + pos] = -1;
        A.occupied[b + pos] = 0;
        const uint4* sd = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
        uint4* dd = reinterpret_cast<uint4*>(A.cur.desc + (b + pos) * 32);
        dd[0] = sd[0]; dd[1] = sd[1];
        // Frame::PosInGrid: posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv)
        const int px = (int)roundf((k.x - 0.f) * C.gw_inv), py = (int)roundf((k.y - 0.f) * C.gh_inv);
        if (px >= 0 && px < PS_GRID_COLS && py >= 0 && py < PS_GRID_ROWS) atomicAdd(&cnt[px * PS_GRID_ROWS + py], 1);
      }
    }
    if (M) { N = kept; st.n = N; }
  }
  if (tid == 0) A.cur.n[s] = N;
  __syncthreads();
  // ---- mGrid as CSR: exclusive scan of the cell counts (12 cells per thread), fill, then every cell's list in keypoint order ----
  {
    static_assert(PS_TRK_NCELL % NTH == 0, "the cell scan takes PS_TRK_NCELL / NTH cells per thread");
    const int per = PS_TRK_NCELL / NTH;   // 12 or 3
    int local[per];
    int sum = 0;
    for (int k = 0; k < per; k++) { local[k] = cnt[tid * per + k]; sum += local[k]; }
    int total;
    int base = block_scan_excl<NTH>(sum, red, &total);
    for (int k = 0; k < per; k++) { cnt[tid * per + k] = base; cursor[tid * per + k] = base; base += local[k]; }
    if (tid == 0) cnt[PS_TRK_NCELL] = total;
  }
  __syncthreads();
  int32_t* coff = A.cur.cell_off + (size_t)s * (PS_TRK_NCELL + 1);
  int32_t* cidx = A.cur.cell_idx + b;
  for (int i = tid; i <= PS_TRK_NCELL; i += NTH) coff[i] = cnt[i];
  for (int i = tid; i < N; i += NTH) {
    const float x = A.cur.x[b + i], y = A.cur.y[b + i];
    const int px = (int)roundf((x - 0.f) * C.gw_inv), py = (int)roundf((y - 0.f) * C.gh_inv);
    if (px >= 0 && px < PS_GRID_COLS && py >= 0 && py < PS_GRID_ROWS) cidx[atomicAdd(&cursor[px * PS_GRID_ROWS + py], 1)] = i;
  }
  __syncthreads();
  for (int c = tid; c < PS_TRK_NCELL; c += NTH) {   // mGrid[x][y].push_back(i) in keypoint order: insertion sort of a short list
    const int b0 = cnt[c], e0 = cnt[c + 1];
    for (int i = b0 + 1; i < e0; i++) {
      const int v = cidx[i];
      int j = i - 1;
      while (j >= b0 && cidx[j] > v) { cidx[j + 1] = cidx[j]; j--; }
      cidx[j + 1] = v;
    }
  }
  __syncthreads();

  if (q.state == TRK_NOT_INITIALIZED) {
    // ---- Tracking::StereoInitialization: more than 500 keypoints; every keypoint with depth becomes a map point ----
    empty_problem(&A.prob_mm1[s], A, s); empty_problem(&A.prob_mm2[s], A, s); empty_problem(&A.prob_lm[s], A, s);
    if (tid == 0) { A.po_vert[s] = PoVertex{(int32_t)b, (int32_t)b}; A.po_prob[s] = PoProb{s, 1, 0, C.fx, C.fy, C.cx, C.cy, C.mbf}; }
    float* tr = A.traj + ((size_t)step * A.S + s) * 16;
    if (N <= 500) {
      if (tid == 0) { q.phase = TRK_PH_IDLE; st.state = TRK_NOT_INITIALIZED; A.stats[(size_t)step * A.S + s] = st; }
      if (tid < 16) tr[tid] = 0.f;
      return;
    }
    float I[16];
    set_identity(I);
    int lm_base = 0;
    for (int i0 = 0; i0 < N; i0 += NTH) {
      const int i = i0 + tid;
      const bool has = i < N && A.cur.depth[b + i] > 0;
      int total;
      const int id = lm_base + block_scan_excl<NTH>(has ? 1 : 0, red, &total);
      if (has) {
        float P[3];
        unproject(C, I, A.cur.x[b + i], A.cur.y[b + i], A.cur.depth[b + i], P);
        for (int c = 0; c < 3; c++) A.cur.xw[3 * (b + i) + c] = P[c];
        A.cur.mp_valid[b + i] = 1; A.cur.mp_observed[b + i] = 1; A.cur.mp_id[b + i] = id;
        // MapPoint::UpdateNormalAndDepth with one observation (MapPoint.cc:470-497): camera centre at the origin
        const float dist = sqrtf(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
        const float maxd = dist * C.sf[A.cur.octave[b + i]];
        for (int c = 0; c < 3; c++) { A.lm_xw[3 * (b + id) + c] = P[c]; A.lm_normal[3 * (b + id) + c] = P[c] / dist; }
        A.lm_maxd[b + id] = maxd; A.lm_mind[b + id] = maxd / C.sf[C.nlevels - 1];
        const uint4* sd = reinterpret_cast<const uint4*>(A.cur.desc + (b + i) * 32);
        uint4* dd = reinterpret_cast<uint4*>(A.lm_desc + (b + id) * 32);
        dd[0] = sd[0]; dd[1] = sd[1];
      }
      lm_base += total;
    }
    if (tid < 16) { A.cur.tcw[s * 16 + tid] = I[tid]; tr[tid] = I[tid]; }
    __syncthreads();
    copy_frame<NTH>(A, A.last, A.cur, s);   // last = F
    if (tid == 0) {
      q.state = TRK_OK; q.have_velocity = 0; q.lm_n = lm_base; q.phase = TRK_PH_IDLE;
      st.state = TRK_OK; st.tracked = 1;
      A.stats[(size_t)step * A.S + s] = st;
    }
    return;
  }

  // ---- Tracking::UpdateLastFrame (localisation mode): the closest keypoints with depth get temporal map points ----
  const int NL = A.last.n[s];
  const float* Tl = A.last.tcw + s * 16;
  for (int i = tid; i < NL; i += NTH) sdepth[i] = A.last.depth[b + i];
  __syncthreads();
  {
    // The reference sorts (depth, index) and walks the list until it has passed 100 points AND the first one beyond
    // 2 * mThDepth: the visited set is every point up to sorted position J = max(c, 100), c = number of near points.  With
    // c >= 100 (any ordinary frame) that is "all near points and the nearest far one" - no ranks needed, one min-reduction;
    // only a frame with fewer than 100 near points needs the positions of its 101 nearest points (rank counting).
    const float thr = 2 * C.th_depth;
    int near = 0;
    unsigned long long far_min = ~0ull;   // (depth bits, index) of the nearest far point: positive floats order like their bit patterns
    for (int i = tid; i < NL; i += NTH) {
      const float d = sdepth[i];
      if (!(d > 0)) continue;
      if (!(d > thr)) near++;
      else { const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i; far_min = k < far_min ? k : far_min; }
    }
    const int c = block_sum_i<NTH>(near, red);
    const int J = c > 100 ? c : 100;
    if (c >= 100) {
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)far_min, dd), hi = __shfl_xor((unsigned)(far_min >> 32), dd);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        far_min = o < far_min ? o : far_min;
      }
      if ((tid & 63) == 0) far_red[tid >> 6] = far_min;
      __syncthreads();
      far_min = far_red[0];
      for (int w = 1; w < NTH / 64; w++) far_min = far_red[w] < far_min ? far_red[w] : far_min;
    }
    const int far_first = (c >= 100 && far_min != ~0ull) ? (int)(unsigned)far_min : -1;
    for (int i0 = 0; i0 < NL; i0 += NTH) {
      const int i = i0 + tid;
      const float d = i < NL ? sdepth[i] : -1.f;
      if (!(d > 0)) continue;
      bool visited;
      if (c >= 100) {
        visited = !(d > thr) || i == far_first;
      } else {
        int rank = 0;   // position in the sort by (depth, index)
        for (int j = 0; j < NL; j++) {
          const float dj = sdepth[j];
          rank += (dj > 0 && (dj < d || (dj == d && j < i))) ? 1 : 0;
        }
        visited = rank <= J;
      }
      if (visited && (!A.last.mp_valid[b + i] || !A.last.mp_observed[b + i])) {
        float P[3];
        unproject(C, Tl, A.last.x[b + i], A.last.y[b + i], d, P);
        for (int k = 0; k < 3; k++) A.last.xw[3 * (b + i) + k] = P[k];
        A.last.mp_valid[b + i] = 1; A.last.mp_observed[b + i] = 0; A.last.mp_id[b + i] = -1;
      }
    }
  }
  __syncthreads();
  // ---- motion model: mCurrentFrame.SetPose(mVelocity * mLastFrame.mTcw); without a velocity the identity (see tracker.py) ----
  if (tid == 0) {
    if (!q.have_velocity) { set_identity(q.velocity); q.have_velocity = 1; }
    float P[16];
    mul4(q.velocity, Tl, P);
    for (int i = 0; i < 16; i++) { pose_pred[i] = P[i]; A.cur.tcw[s * 16 + i] = P[i]; }
  }
  for (int i = tid; i < NL; i += NTH) A.qvalid[b + i] = (A.last.mp_valid[b + i] && !A.last.outlier[b + i]) ? 1 : 0;
  __syncthreads();
  if (tid == 0) {
    // matcher(0.9, true).SearchByProjection(mCurrentFrame, mLastFrame, th, false) (Tracking.cc:3030-3048), th = 7 then 14
    PjProb d;
    for (int i = 0; i < (int)(sizeof(PjProb) / 4); i++) ((int32_t*)&d)[i] = 0;
    fill_search_common(d, A, s, N, NL);
    d.th_dist = 100; d.ratio_test = 0; d.nn_ratio = 0.9f; d.check_ori = 1; d.frame_mode = 1;
    for (int i = 0; i < 16; i++) { d.tcw[i] = pose_pred[i]; d.tlw[i] = Tl[i]; }
    d.th = 7.f;
    A.prob_mm1[s] = d;
    d.th = 14.f;
    A.prob_mm2[s] = d;
    empty_problem(&A.prob_lm[s], A, s);
    A.po_vert[s] = PoVertex{(int32_t)b, (int32_t)b};
    A.po_prob[s] = PoProb{s, 1, 0, C.fx, C.fy, C.cx, C.cy, C.mbf};
    q.phase = TRK_PH_MM; q.retried = 0; q.lm_searched = 0; q.nvalid_pose = 0;
    A.stats[(size_t)step * A.S + s] = st;
  }
}

// Step 2: `if (nmatches < 20) retry with 2 * th` (Tracking.cc:3042-3048)
template <int NTH>
__global__ __launch_bounds__(NTH) void trk_after_mm1(TrkArrays A) {
  const int s = blockIdx.x, tid = threadIdx.x;
  TrkSeq& q = A.seq[s];
  if (q.phase != TRK_PH_MM) return;
  const size_t b = (size_t)s * A.cap;
  if (A.nmatch_mm1[s] < 20) {
    const int NL = A.last.n[s];
    for (int i = tid; i < NL; i += NTH) A.qvalid[b + i] = (A.last.mp_valid[b + i] && !A.last.outlier[b + i]) ? 1 : 0;   // pj_project cleared the misses
    if (tid == 0) q.retried = 1;
  } else if (tid == 0) {
    A.prob_mm2[s].nq = 0; A.prob_mm2[s].nt = 0;
  }
}

// Optimizer::PoseOptimization(&mCurrentFrame) problem of sequence s (OdoSequence::fillPose)
template <int NTH>
__device__ void fill_pose(const TrkArrays& A, int s, int N, int* red) {
  const int tid = threadIdx.x;
  const size_t b = (size_t)s * A.cap;
  int nv = 0;
  for (int i = tid; i < N; i += NTH) {
    A.po_obs[3 * (b + i)] = A.cur.x[b + i]; A.po_obs[3 * (b + i) + 1] = A.cur.y[b + i]; A.po_obs[3 * (b + i) + 2] = A.cur.uright[b + i];
    A.po_is2[b + i] = A.cam.inv_sigma2[A.cur.octave[b + i]];
    nv += A.cur.mp_valid[b + i] ? 1 : 0;
  }
  nv = block_sum_i<NTH>(nv, red);
  if (tid == 0) {
    A.seq[s].nvalid_pose = nv;
    const Se3 T = se3_from_mat4f(A.cur.tcw + s * 16);   // Converter::toSE3Quat(pFrame->mTcw)
    double* p = A.po_pose + (size_t)s * 7;
    p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2]; p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
    A.po_vert[s] = PoVertex{(int32_t)b, (int32_t)(b + N)};
  }
}

// pFrame->SetPose(...) unless the optimiser returned before it (fewer than 15 correspondences, Optimizer.cc:376-377)
__device__ void take_pose(const TrkArrays& A, int s) {
  if (threadIdx.x == 0 && A.seq[s].nvalid_pose >= 15) {
    const double* p = A.po_pose + (size_t)s * 7;
    Se3 T;
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
    se3_to_mat4f(T, A.cur.tcw + s * 16);
  }
  __syncthreads();
}

// Step 3: the matches of the motion-model search become the frame's map points (Tracking.cc:3050-3056), then the pose problem
template <int NTH>
__global__ __launch_bounds__(NTH) void trk_after_mm(TrkArrays A, int step) {
  __shared__ int red[NTH / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  TrkSeq& q = A.seq[s];
  if (q.phase != TRK_PH_MM) return;
  const size_t b = (size_t)s * A.cap;
  TrkStat st = A.stats[(size_t)step * A.S + s];
  const int nm = q.retried ? A.nmatch_mm2[s] : A.nmatch_mm1[s];
  st.mm_matches = nm; st.retried = q.retried;
  __syncthreads();
  if (nm < 20) { if (tid == 0) seq_fail(A, s, step, st); return; }
  const int N = A.cur.n[s];
  for (int j = tid; j < N; j += NTH) {
    const int i = A.match[b + j];
    A.cur.mp_valid[b + j] = i >= 0;
    if (i >= 0) {
      for (int c = 0; c < 3; c++) A.cur.xw[3 * (b + j) + c] = A.last.xw[3 * (b + i) + c];
      A.cur.mp_observed[b + j] = A.last.mp_observed[b + i]; A.cur.mp_id[b + j] = A.last.mp_id[b + i];
    }
  }
  __syncthreads();
  fill_pose<NTH>(A, s, N, red);
  if (tid == 0) { q.phase = TRK_PH_POSE1; A.stats[(size_t)step * A.S + s] = st; }
}

// Step 4: discard outliers (Tracking.cc:3062-3082), then SearchLocalPoints: Frame::isInFrustum (Frame.cc:1686-1743) +
// the SearchByProjection(mCurrentFrame, points, th = 1) problem with matcher(0.8) (Tracking.cc:3097-3160)
template <int NTH>
__global__ __launch_bounds__(NTH) void trk_after_pose1(TrkArrays A, int step) {
  __shared__ int red[NTH / 64];
  __shared__ uint8_t already[4352];
  const int s = blockIdx.x, tid = threadIdx.x;
  TrkSeq& q = A.seq[s];
  if (q.phase != TRK_PH_POSE1) return;
  const TrkCam& C = A.cam;
  const size_t b = (size_t)s * A.cap;
  TrkStat st = A.stats[(size_t)step * A.S + s];
  take_pose(A, s);
  const int N = A.cur.n[s], n = q.lm_n;
  for (int i = tid; i < n; i += NTH) already[i] = 0;
  __syncthreads();
  int nmatches = 0, nmap = 0;
  for (int i = tid; i < N; i += NTH) {
    if (!A.cur.mp_valid[b + i]) continue;
    if (A.cur.outlier[b + i]) { A.cur.mp_valid[b + i] = 0; A.cur.outlier[b + i] = 0; continue; }
    nmatches++;
    if (A.cur.mp_observed[b + i]) nmap++;
  }
  nmatches = block_sum_i<NTH>(nmatches, red);
  nmap = block_sum_i<NTH>(nmap, red);
  st.matches = nmatches; st.map_matches = nmap;
  if (!(nmatches > 20)) { if (tid == 0) seq_fail(A, s, step, st); return; }
  if (nmap < 10) {   // mbVO: the frame is kept without TrackLocalMap
    // (no second PoseOptimization for this frame: an empty problem, or the launch after TrackLocalMap's search would optimise the
    // frame again from the optimised pose and rewrite its outlier flags - the host loop and the reference do not)
    if (tid == 0) { q.phase = TRK_PH_FINISH; A.po_vert[s] = PoVertex{(int32_t)b, (int32_t)b}; A.stats[(size_t)step * A.S + s] = st; }
    return;
  }
  for (int i = tid; i < N; i += NTH) {
    const bool v = A.cur.mp_valid[b + i];
    if (A.cur.mp_id[b + i] >= 0) already[A.cur.mp_id[b + i]] = 1;   // mnLastFrameSeen: matched points and the outliers just discarded (Tracking.cc:3071-3075)
    A.occupied[b + i] = (v && A.cur.mp_observed[b + i]) ? 1 : 0;
  }
  __syncthreads();
  const float* T = A.cur.tcw + s * 16;
  float Ow[3];
  for (int r = 0; r < 3; r++) Ow[r] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
  int nto = 0;
  for (int i = tid; i < n; i += NTH) {
    A.qvalid[b + i] = 0; A.qu[b + i] = 0.f; A.qv[b + i] = 0.f; A.qur[b + i] = 0.f; A.qrad[b + i] = 0.f; A.qminl[b + i] = 0; A.qmaxl[b + i] = 0;
    if (already[i]) continue;
    const float* P = A.lm_xw + 3 * (b + i);
    const float PcX = T[0] * P[0] + T[1] * P[1] + T[2] * P[2] + T[3], PcY = T[4] * P[0] + T[5] * P[1] + T[6] * P[2] + T[7],
                PcZ = T[8] * P[0] + T[9] * P[1] + T[10] * P[2] + T[11];
    if (PcZ < 0.0f) continue;
    const float invz = 1.0f / PcZ, u = C.fx * PcX * invz + C.cx, v = C.fy * PcY * invz + C.cy;
    if (u < 0 || u > (float)C.w || v < 0 || v > (float)C.h) continue;
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const float dist = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    const float maxd = A.lm_maxd[b + i];
    if (dist < 0.8f * A.lm_mind[b + i] || dist > 1.2f * maxd) continue;
    const float* Pn = A.lm_normal + 3 * (b + i);
    const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
    if (viewCos < 0.5f) continue;
    // MapPoint::PredictScale: ceil(log(mfMaxDistance / currentDist) / mfLogScaleFactor); the float logarithm is taken as the
    // rounded double one (the host's logf is within one ulp of it; a level only moves when the quotient sits on an integer)
    int level = (int)ceilf((float)log((double)(maxd / dist)) / C.log_sf);
    level = level < 0 ? 0 : (level >= C.nlevels ? C.nlevels - 1 : level);
    const float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;   // ORBmatcher::RadiusByViewingCos, th = 1
    A.qvalid[b + i] = 1; A.qu[b + i] = u; A.qv[b + i] = v; A.qur[b + i] = u - C.mbf * invz; A.qrad[b + i] = r * C.sf[level];
    A.qminl[b + i] = level - 1; A.qmaxl[b + i] = level;
    nto++;
  }
  nto = block_sum_i<NTH>(nto, red);
  st.lm_candidates = nto;
  if (nto > 0) {
    if (tid == 0) {
      PjProb d;
      for (int i = 0; i < (int)(sizeof(PjProb) / 4); i++) ((int32_t*)&d)[i] = 0;
      fill_search_common(d, A, s, N, n);
      d.th_dist = 100; d.ratio_test = 1; d.nn_ratio = 0.8f; d.check_ori = 0; d.frame_mode = 0; d.th = 1.f;
      A.prob_lm[s] = d;
      q.lm_searched = 1; q.phase = TRK_PH_LM;
      A.stats[(size_t)step * A.S + s] = st;
    }
    return;
  }
  fill_pose<NTH>(A, s, N, red);
  if (tid == 0) { q.lm_searched = 0; q.phase = TRK_PH_POSE2; A.stats[(size_t)step * A.S + s] = st; }
}

// Step 5: the local-map matches join the frame's map points, then the second pose problem
template <int NTH>
__global__ __launch_bounds__(NTH) void trk_after_lm(TrkArrays A) {
  __shared__ int red[NTH / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  TrkSeq& q = A.seq[s];
  if (q.phase != TRK_PH_LM) return;
  const size_t b = (size_t)s * A.cap;
  const int N = A.cur.n[s];
  for (int j = tid; j < N; j += NTH) {
    const int i = A.match[b + j];
    if (i < 0) continue;
    A.cur.mp_valid[b + j] = 1; A.cur.mp_observed[b + j] = 1; A.cur.mp_id[b + j] = i;
    for (int c = 0; c < 3; c++) A.cur.xw[3 * (b + j) + c] = A.lm_xw[3 * (b + i) + c];
  }
  __syncthreads();
  fill_pose<NTH>(A, s, N, red);
  if (tid == 0) q.phase = TRK_PH_POSE2;
}

// Step 6: TrackLocalMap's inlier count (Tracking.cc:3128-3158), the motion model (Tracking.cc:1260-1286), last = current
template <int NTH>
__global__ __launch_bounds__(NTH) void trk_finish(TrkArrays A, int step) {
  __shared__ int red[NTH / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  TrkSeq& q = A.seq[s];
  const int ph = q.phase;
  if (ph != TRK_PH_POSE2 && ph != TRK_PH_FINISH) return;
  const size_t b = (size_t)s * A.cap;
  TrkStat st = A.stats[(size_t)step * A.S + s];
  const int N = A.cur.n[s];
  __syncthreads();
  if (ph == TRK_PH_POSE2) {
    take_pose(A, s);
    int inl = 0;
    for (int i = tid; i < N; i += NTH) {
      if (!A.cur.mp_valid[b + i]) continue;
      if (A.cur.outlier[b + i]) A.cur.mp_valid[b + i] = 0;   // stereo: outliers lose their map point (Tracking.cc:3141-3142)
      else inl++;
    }
    inl = block_sum_i<NTH>(inl, red);
    st.lm_inliers = inl;
    if (inl < 30) { if (tid == 0) seq_fail(A, s, step, st); return; }
  }
  // clean VO matches (Tracking.cc:1274-1286)
  for (int i = tid; i < N; i += NTH)
    if (A.cur.mp_valid[b + i] && !A.cur.mp_observed[b + i]) { A.cur.mp_valid[b + i] = 0; A.cur.outlier[b + i] = 0; }
  if (tid == 0) {
    // mVelocity = mCurrentFrame.mTcw * LastTwc (Tracking.cc:1260-1270)
    const float* L = A.last.tcw + s * 16;
    const float* F = A.cur.tcw + s * 16;
    float lastTwc[16];
    set_identity(lastTwc);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) lastTwc[4 * r + c] = L[4 * c + r];
    for (int r = 0; r < 3; r++) {
      float acc = 0;
      for (int c = 0; c < 3; c++) acc += L[4 * c + r] * L[4 * c + 3];
      lastTwc[4 * r + 3] = -acc;
    }
    float V[16];
    mul4(F, lastTwc, V);
    for (int i = 0; i < 16; i++) q.velocity[i] = V[i];
    float* tr = A.traj + ((size_t)step * A.S + s) * 16;
    for (int i = 0; i < 16; i++) tr[i] = F[i];
  }
  __syncthreads();
  copy_frame<NTH>(A, A.last, A.cur, s);
  if (tid == 0) {
    q.phase = TRK_PH_IDLE; q.state = TRK_OK;   // `if (bOK) mState = OK;` - also after a frame that was lost
    st.state = TRK_OK; st.tracked = 1;
    A.stats[(size_t)step * A.S + s] = st;
  }
}

// the search windows of this step that held more candidates than the matcher stores (PS_PJ_CAP), per sequence: stamped into the
// step's statistics, counter cleared for the next step
__global__ __launch_bounds__(256) void trk_stamp_overflow(TrkArrays A, int32_t* overflow, int step) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= A.S) return;
  A.stats[(size_t)step * A.S + s].pad[0] = overflow[s];
  overflow[s] = 0;
}

}  // namespace

extern "C" {
void psk_trk_stamp_overflow(const TrkArrays* A, int32_t* overflow, int step, hipStream_t st) {
  hipLaunchKernelGGL(trk_stamp_overflow, dim3((A->S + 255) / 256), dim3(256), 0, st, *A, overflow, step);
}
static inline bool trk_small(const TrkArrays* A) { return A->S <= TRK_SMALL_S; }
void psk_trk_begin(const TrkArrays* A, int step, hipStream_t st) {
  if (trk_small(A)) hipLaunchKernelGGL(trk_begin<TRK_T_SMALL>, dim3(A->S), dim3(TRK_T_SMALL), 0, st, *A, step);
  else hipLaunchKernelGGL(trk_begin<TRK_T>, dim3(A->S), dim3(TRK_T), 0, st, *A, step);
}
void psk_trk_after_mm1(const TrkArrays* A, hipStream_t st) {
  if (trk_small(A)) hipLaunchKernelGGL(trk_after_mm1<TRK_T_SMALL>, dim3(A->S), dim3(TRK_T_SMALL), 0, st, *A);
  else hipLaunchKernelGGL(trk_after_mm1<TRK_T>, dim3(A->S), dim3(TRK_T), 0, st, *A);
}
void psk_trk_after_mm(const TrkArrays* A, int step, hipStream_t st) {
  if (trk_small(A)) hipLaunchKernelGGL(trk_after_mm<TRK_T_SMALL>, dim3(A->S), dim3(TRK_T_SMALL), 0, st, *A, step);
  else hipLaunchKernelGGL(trk_after_mm<TRK_T>, dim3(A->S), dim3(TRK_T), 0, st, *A, step);
}
void psk_trk_after_pose1(const TrkArrays* A, int step, hipStream_t st) {
  if (trk_small(A)) hipLaunchKernelGGL(trk_after_pose1<TRK_T_SMALL>, dim3(A->S), dim3(TRK_T_SMALL), 0, st, *A, step);
  else hipLaunchKernelGGL(trk_after_pose1<TRK_T>, dim3(A->S), dim3(TRK_T), 0, st, *A, step);
}
void psk_trk_after_lm(const TrkArrays* A, hipStream_t st) {
  if (trk_small(A)) hipLaunchKernelGGL(trk_after_lm<TRK_T_SMALL>, dim3(A->S), dim3(TRK_T_SMALL), 0, st, *A);
  else hipLaunchKernelGGL(trk_after_lm<TRK_T>, dim3(A->S), dim3(TRK_T), 0, st, *A);
}
void psk_trk_finish(const TrkArrays* A, int step, hipStream_t st) {
  if (trk_small(A)) hipLaunchKernelGGL(trk_finish<TRK_T_SMALL>, dim3(A->S), dim3(TRK_T_SMALL), 0, st, *A, step);
  else hipLaunchKernelGGL(trk_finish<TRK_T>, dim3(A->S), dim3(TRK_T), 0, st, *A, step);
}
}
