// Glue kernels of the object half of the device-resident lockstep tracker: the host side of Tracking::Track's object functions
// between the hot-path calls, moved onto the device so that a frame's object chain
//   ExtractObjORB (cvb_*) -> ComputeObjStereoMatches (st_*) -> AssignFeatures -> TrackMapObject -> SearchByBruceMatching (bf_*) ->
//   CFSE3ObjStateOptimization (pose_lm) -> SearchObjectLocalPoints / SearchByProjection(F, nOrder, MOPs) (pj_*) -> CFSE3 -> end of Track
// is one stream of launches behind the camera chain of the same step, with no host round trip.  The per-call twin
// pointslot_amd/object_tracker.py documents the slice and the mapping to /root/reference/src/Tracking.cc:1224-1233,1443-1478,
// 1533-2031,2288-2712 and src/Frame.cc:690-733,762-977,1744-1806; every float / double expression here is written operation by
// operation as that file evaluates it (this file is compiled with -ffp-contract=off) - tests/test_object_device_gpu.py compares the
// two drivers bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pointslot_hip.h"
#include "objtrack_plan.h"
#include "se3.h"

namespace {

#define OB_T 256

__device__ __forceinline__ int ob_block_sum(int v, int* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < OB_T / 64; w++) t += red[w];
  return t;
}
__device__ __forceinline__ int ob_block_scan_excl(int v, int* red, int* total) {
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
  for (int w = 0; w < OB_T / 64; w++) t += red[w];
  *total = t;
  return base + incl - v;
}

__device__ __forceinline__ Se3 ob_pose(const double* p) {
  Se3 T;
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
  return T;
}
__device__ __forceinline__ void ob_store_pose(double* p, const Se3& T) {
  p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2]; p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
}

// the camera tracker's view of the frame: was the tracker initialised before it, and the poses of this and the last frame
__device__ __forceinline__ bool ob_camera_initialized(const ObArrays& A, int s, int step) {
  return step > 0 && A.cam_stats[((size_t)(step - 1) * A.S + s) * A.cam_stat_words] != 0;   // TrkStat::state after the frame before
}
__device__ __forceinline__ const float* ob_cam_pose(const ObArrays& A, int s, int step) {      // nullptr: the frame has no pose
  if (step < 0) return nullptr;
  if (A.cam_stats[((size_t)step * A.S + s) * A.cam_stat_words + 1] == 0) return nullptr;       // TrkStat::tracked
  return A.cam_traj + ((size_t)step * A.S + s) * 16;
}

// ---------------------------------------------------------------------------------------------------------------------
// LeftObjMask / RightObjMask of Frame::ExtractObjORB (Frame.cc:2632-2643) from the left 8-bit id mask; the right id mask is
// Frame::ReadKittiSegmentationImage(.., rightseg = true) (Frame.cc:1217-1290): scanning a row from the left, every labelled
// pixel writes its label 49 pixels to both sides, so a pixel ends up with the label of the RIGHTMOST labelled pixel within
// (x, x + 49] or else with its own (column 0 is never written from the right).  One workgroup per image row.
// ---------------------------------------------------------------------------------------------------------------------
// ONE WAVE PER ROW, one row per workgroup: the row and its `run` table live in LDS, the prefix maximum is a wave scan - no workgroup
// barrier anywhere.  The detector's 8 x 8 occupancy cells, which this kernel fills on the way (occ[image][cell], cleared by
// psi_cvorb_batch_begin), get a 1 from every row that has a mask pixel in them.  (Until r04 eight rows shared a 512-thread workgroup
// with 31 KB of LDS so that they could publish a cell row together; beside the other lockstep groups' kernels such a workgroup
// waited for its place: 527 us per launch against 103 alone.)
#define OB_MASK_T 64
__global__ __launch_bounds__(OB_MASK_T) void ob_masks(ObArrays A, uint8_t* objmask, int W, int H, int ostride, uint8_t* occ, int ocw, int och) {
  extern __shared__ __attribute__((aligned(16))) uint8_t row_all[];
  const int WP = (W + 255) & ~255, per = WP >> 6;             // pixels per lane: a multiple of 4
  const int s = blockIdx.y, lane = threadIdx.x;
  uint8_t* row_sm = row_all;
  int16_t* run = reinterpret_cast<int16_t*>(row_sm + WP);
  const int y = blockIdx.x;
  uint8_t* occL = occ + ((size_t)(2 * s) * och + (y >> 3)) * ocw;
  uint8_t* occR = occ + ((size_t)(2 * s + 1) * och + (y >> 3)) * ocw;
  {
    const uint8_t* M = A.idmask + (size_t)s * A.mask_pitch + (size_t)y * A.mask_stride;
    // the lane's pixels [c0, c0 + per) as (unaligned) dwords into LDS; its rightmost labelled column on the way
    const int c0 = lane * per;
    int local = -1;
    for (int q0 = 0; q0 < per; q0 += 32) {             // up to eight dwords of the row per lane and step, requested together
      uint32_t vv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int x = c0 + q0 + 4 * u;
        uint32_t v = 0;
        if (q0 + 4 * u < per) {
          if (x + 3 < W) __builtin_memcpy(&v, M + x, 4);
          else
            for (int j = 0; j < 4; j++) if (x + j < W) v |= (uint32_t)M[x + j] << (8 * j);
        }
        vv[u] = v;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int x = c0 + q0 + 4 * u;
        if (q0 + 4 * u >= per) continue;
        *reinterpret_cast<uint32_t*>(row_sm + x) = vv[u];
        if (vv[u]) local = x + 3 - (__builtin_clz(vv[u]) >> 3);
      }
    }
    // a row without a labelled pixel (most rows: the objects cover a few percent of the image) leaves two rows of zeros and no cell
    // (r05: the kernel is bound by vector instruction issue - 1.0 busy in tools/valu_busy.sh - and nine tenths of them are the
    // per-pixel passes below)
    if (!__any(local >= 0)) {
      uint8_t* L0 = objmask + ((size_t)(2 * s) * H + y) * ostride;
      uint8_t* R0 = objmask + ((size_t)(2 * s + 1) * H + y) * ostride;
      for (int x4 = lane * 4; x4 < ostride; x4 += 256) { *reinterpret_cast<uint32_t*>(L0 + x4) = 0u; *reinterpret_cast<uint32_t*>(R0 + x4) = 0u; }
      return;
    }
    // run[c] = rightmost labelled column <= c (-1: none): the exclusive prefix maximum over the lanes, then the lane's own pixels
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl = max(incl, o); }
    int r = __shfl_up(incl, 1);
    if (lane == 0) r = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int q = 0; q < per; q += 4) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(row_sm + c0 + q);
      int16_t rr[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { if ((v >> (8 * j)) & 0xFFu) r = c0 + q + j; rr[j] = (int16_t)r; }
      *reinterpret_cast<uint2*>(run + c0 + q) = make_uint2((uint32_t)(uint16_t)rr[0] | ((uint32_t)(uint16_t)rr[1] << 16), (uint32_t)(uint16_t)rr[2] | ((uint32_t)(uint16_t)rr[3] << 16));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t* L = objmask + ((size_t)(2 * s) * H + y) * ostride;
    uint8_t* R = objmask + ((size_t)(2 * s + 1) * H + y) * ostride;
    for (int x4 = lane * 4; x4 < ostride; x4 += 256) {       // four pixels per lane, one aligned dword per mask (columns beyond W: 0)
      uint32_t lv = 0, rv = 0;
      const uint32_t m4 = *reinterpret_cast<const uint32_t*>(row_sm + x4);     // bytes beyond W are zero
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int x = x4 + j;
        if (x >= W) break;
        const uint32_t m = (m4 >> (8 * j)) & 0xFFu;
        const uint32_t lb = (m != 0 && m != 255) ? 255u : 0u;
        const int rr = run[min(x + 49, W - 1)];
        const uint32_t lab = (rr > x && x > 0) ? (uint32_t)row_sm[rr] : m;
        const uint32_t rb = (lab != 0 && lab != 255) ? 255u : 0u;
        lv |= lb << (8 * j); rv |= rb << (8 * j);
      }
      *reinterpret_cast<uint32_t*>(L + x4) = lv;
      *reinterpret_cast<uint32_t*>(R + x4) = rv;
      if (lv) occL[x4 >> 3] = 1;
      if (rv) occR[x4 >> 3] = 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Frame::Frame, object part after ExtractObjORB / ComputeObjStereoMatches: AssignFeatures (Frame.cc:762-977; the temp object
// keys go to the detection whose id is the mask label - 1), AssignDetObjFeasToGrid (:1863-1888), and the lookup of the
// detections' MapObjects (AllObjects by mnTruthID, Tracking.cc:1548-1551).  One workgroup per sequence.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OB_T) void ob_begin(ObArrays A, int step) {
  __shared__ int cnt[OB_NCELL + 1];
  __shared__ int cursor[OB_NCELL];
  __shared__ int red[OB_T / 64];
  __shared__ int s_off[OB_MAXK + 1];
  const int s = blockIdx.x, tid = threadIdx.x, K = A.K;
  const ObCam& C = A.cam;
  const size_t fb = (size_t)s * A.OC;
  ObFrame& F = A.cur;
  int ndet = 0;
  for (int j = 0; j < K; j++) if (A.dets_in[(size_t)s * K + j].id >= 0) ndet = j + 1;
  if (tid < K) {
    F.det[(size_t)s * K + tid] = A.dets_in[(size_t)s * K + tid];
    F.mo[(size_t)s * K + tid] = -1;
    A.in_last[(size_t)s * K + tid] = -1; A.tracked[(size_t)s * K + tid] = 0; A.need[(size_t)s * K + tid] = 0; A.track_ok[(size_t)s * K + tid] = 0;
  }
  if (tid == 0) F.ndet[s] = ndet;
  const ps_keypoint* kps = (const ps_keypoint*)A.cv_kps + (size_t)(2 * s) * A.cv_cap;
  const uint8_t* desc = A.cv_desc + (size_t)(2 * s) * A.cv_cap * 32;
  int N = A.cv_count[2 * s];
  N = N < A.cv_cap ? N : A.cv_cap;
  const uint8_t* M = A.idmask + (size_t)s * A.mask_pitch;
  int8_t* owner = A.owner + (size_t)s * A.cv_cap;
  for (int i = tid; i < N; i += OB_T) {
    const ps_keypoint k = kps[i];
    const int lab = M[(size_t)(int)k.y * A.mask_stride + (int)k.x];
    int own = -1;
    if (lab != 0 && lab != 255)
      for (int j = 0; j < ndet; j++) {
        const int id = A.dets_in[(size_t)s * K + j].id;
        if ((id > 255 ? id - 255 : id) == lab - 1) { own = j; break; }
      }
    owner[i] = (int8_t)own;
  }
  __syncthreads();
  int run = 0;
  for (int j = 0; j < K; j++) {
    if (tid == 0) s_off[j] = run;
    if (j >= ndet) continue;
    for (int i0 = 0; i0 < N; i0 += OB_T) {
      const int i = i0 + tid;
      const bool mine = i < N && owner[i] == j;
      int total;
      const int pos = run + ob_block_scan_excl(mine ? 1 : 0, red, &total);
      if (mine && pos < A.OC) {
        const ps_keypoint k = kps[i];
        const size_t o = fb + pos;
        F.x[o] = k.x; F.y[o] = k.y; F.angle[o] = k.angle; F.octave[o] = k.octave;
        F.uright[o] = A.st_uright[fb + i]; F.depth[o] = A.st_depth[fb + i];
        const uint4* sd = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
        uint4* dd = reinterpret_cast<uint4*>(F.desc + o * 32);
        dd[0] = sd[0]; dd[1] = sd[1];
        F.mp_valid[o] = 0; F.mp_observed[o] = 0; F.outlier[o] = 0; F.mp_id[o] = -1;
        F.mp_po[3 * o] = 0.f; F.mp_po[3 * o + 1] = 0.f; F.mp_po[3 * o + 2] = 0.f;
        A.occupied[o] = 0; A.inbbox[o] = 0;
      }
      run += total;
    }
  }
  run = run < A.OC ? run : A.OC;
  if (tid == 0) { s_off[K] = run; for (int j = ndet; j < K; j++) s_off[j] = run; }
  __syncthreads();
  if (tid <= K) F.off[(size_t)s * (K + 1) + tid] = s_off[tid];
  // ---- mvObjKeysGrid of every detection as CSR (the 64 x 48 grid over the image, Frame::PosInGrid) ----
  const bool init = ob_camera_initialized(A, s, step);
  for (int j = 0; j < ndet; j++) {
    const int b0 = s_off[j], n = s_off[j + 1] - b0;
    for (int i = tid; i <= OB_NCELL; i += OB_T) cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += OB_T) {
      const int px = (int)roundf((F.x[fb + b0 + i] - 0.f) * C.gw_inv), py = (int)roundf((F.y[fb + b0 + i] - 0.f) * C.gh_inv);
      if (px >= 0 && px < PS_GRID_COLS && py >= 0 && py < PS_GRID_ROWS) atomicAdd(&cnt[px * PS_GRID_ROWS + py], 1);
    }
    __syncthreads();
    {
      const int per = OB_NCELL / OB_T;
      int local[per];
      int sum = 0;
      for (int k = 0; k < per; k++) { local[k] = cnt[tid * per + k]; sum += local[k]; }
      int total;
      int base = ob_block_scan_excl(sum, red, &total);
      for (int k = 0; k < per; k++) { cnt[tid * per + k] = base; cursor[tid * per + k] = base; base += local[k]; }
      if (tid == 0) cnt[OB_NCELL] = total;
    }
    __syncthreads();
    int32_t* coff = F.cell_off + ((size_t)s * K + j) * (OB_NCELL + 1);
    int32_t* cidx = F.cell_idx + fb + b0;
    for (int i = tid; i <= OB_NCELL; i += OB_T) coff[i] = cnt[i];
    for (int i = tid; i < n; i += OB_T) {
      const int px = (int)roundf((F.x[fb + b0 + i] - 0.f) * C.gw_inv), py = (int)roundf((F.y[fb + b0 + i] - 0.f) * C.gh_inv);
      if (px >= 0 && px < PS_GRID_COLS && py >= 0 && py < PS_GRID_ROWS) cidx[atomicAdd(&cursor[px * PS_GRID_ROWS + py], 1)] = i;
    }
    __syncthreads();
    for (int c = tid; c < OB_NCELL; c += OB_T) {   // push_back order: ascending feature index inside a cell
      const int a = cnt[c], e = cnt[c + 1];
      for (int i = a + 1; i < e; i++) {
        const int v = cidx[i];
        int k = i - 1;
        while (k >= a && cidx[k] > v) { cidx[k + 1] = cidx[k]; k--; }
        cidx[k + 1] = v;
      }
    }
    __syncthreads();
  }
  // ---- statistics, MapObjects of the detections; first observations reserve a free slot in detection order ----
  for (int j = 0; j < ndet; j++) {
    const int b0 = s_off[j], n = s_off[j + 1] - b0;
    int st = 0;
    for (int i = tid; i < n; i += OB_T) st += F.depth[fb + b0 + i] > 0 ? 1 : 0;
    st = ob_block_sum(st, red);
    if (tid == 0) {
      ObStat o;
      for (int i = 0; i < (int)(sizeof(ObStat) / 4); i++) ((int32_t*)&o)[i] = 0;
      o.id = F.det[(size_t)s * K + j].id; o.n = n; o.stereo = st;
      o.dynamic = 1; o.mo_dynamic = -1;      // a DetectionObject is born dynamic (DetectionObject.cc:74)
      A.stats[((size_t)step * A.S + s) * K + j] = o;
    }
  }
  for (int j = ndet + tid; j < K; j += OB_T) {
    ObStat o;
    for (int i = 0; i < (int)(sizeof(ObStat) / 4); i++) ((int32_t*)&o)[i] = 0;
    o.id = -1;
    A.stats[((size_t)step * A.S + s) * K + j] = o;
  }
  // the detector's capacity flags live in the area it clears before every batch: remembered here, per sequence, for every step
  if (tid == 0 && (A.cv_overflow[2 * s] | A.cv_overflow[2 * s + 1])) A.det_overflow[s]++;
  if (tid == 0 && init) {
    ObMapObject* T = A.mobj + (size_t)s * A.M;
    unsigned long long reserved = 0;
    for (int j = 0; j < ndet; j++) {
      const int id = F.det[(size_t)s * K + j].id;
      int slot = -1;
      // an unused slot (id < 0) in front of the last used one is no detection: no MapObject (free table slots carry id -1 too)
      if (id < 0) { F.mo[(size_t)s * K + j] = -1; continue; }
      for (int m = 0; m < A.M; m++) if (T[m].id == id) { slot = m; break; }
      if (slot < 0) {
        int fr = -1;
        for (int m = 0; m < A.M; m++) if (T[m].id < 0 && !((reserved >> m) & 1ull)) { fr = m; break; }
        if (fr < 0) { A.dropped[s]++; slot = -1; }
        else { reserved |= 1ull << fr; slot = -2 - fr; }     // reserved for MapObjectInit
      }
      F.mo[(size_t)s * K + j] = slot;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Device functions of one detection's workgroup
// ---------------------------------------------------------------------------------------------------------------------
struct ObShared {
  int red[OB_T / 64];
  unsigned long long red64[OB_T / 64];
  double mean[3];
  int ibuf[8];
};

// the detection's camera-frame points in `order` (feature indices relative to the detection, or nullptr = every feature with depth,
// ascending): Frame::UnprojectStereodynamic (Frame.cc:2521-2544), float arithmetic, widened to double
__device__ void ob_cam_points(const ObArrays& A, const ObFrame& F, size_t fo, int n, double* pts, int32_t* pidx, int* l_out, ObShared& sh) {
  const ObCam& C = A.cam;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += OB_T) {
    const int i = i0 + threadIdx.x;
    const bool has = i < n && F.depth[fo + i] > 0;
    int total;
    const int u = base + ob_block_scan_excl(has ? 1 : 0, sh.red, &total);
    if (has) {
      const float z = F.depth[fo + i];
      const float xx = (F.x[fo + i] - C.cx) * z * C.inv_fx, yy = (F.y[fo + i] - C.cy) * z * C.inv_fy;
      pts[3 * u] = (double)xx; pts[3 * u + 1] = (double)yy; pts[3 * u + 2] = (double)z;
      pidx[u] = i;
    }
    base += total;
  }
  __syncthreads();
  *l_out = base;
}

// the RANSAC centroid of InitializeCurrentObjPose / MapObjectInit / MapObjectReInit (Tracking.cc:1656-1700): `iterations` draws
// of cv::RNG (default state), score = points within fmax of the drawn one, the first best draw's inliers (flags, ascending) and
// their mean summed in that order.  score / draw scratch: scr[0 .. iterations) ints.  Returns the inlier count.
// (the points and the inlier flags of the usual detection - up to OB_RANSAC_LDS stereo points - are worked on in LDS: every draw walks
// all points, and the centroid is one lane's sequential sum, in the reference's order)
#define OB_RANSAC_LDS 384
__device__ int ob_ransac(const double* pts_g, int l, float fmax, int iterations, int32_t* scr, uint8_t* flag_g, double* mean, ObShared& sh) {
  __shared__ double pl[3 * OB_RANSAC_LDS];
  __shared__ uint8_t fl[OB_RANSAC_LDS];
  const int tid = threadIdx.x;
  if (l == 0 || iterations <= 0) { if (tid < 3) mean[tid] = 0; __syncthreads(); return 0; }
  const bool in_lds = l <= OB_RANSAC_LDS;
  if (in_lds)
    for (int i = tid; i < 3 * l; i += OB_T) pl[i] = pts_g[i];
  const double* pts = in_lds ? pl : pts_g;
  // cv::RNG's multiply-with-carry stream is a serial recurrence; only the recurrence runs on one lane, the reduction of a state to
  // a point index (RNG::operator()(unsigned): next() % l) is done by the draws' own threads
  if (tid == 0) {
    unsigned long long state = 0xFFFFFFFFull;
    for (int k = 0; k < iterations; k++) {
      state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32);
      scr[k] = (int)(unsigned)state;
    }
  }
  __syncthreads();
  for (int k = tid; k < iterations; k += OB_T) scr[k] = (int)((unsigned)scr[k] % (unsigned)l);
  __syncthreads();
  const double fm = (double)fmax;
  // "sqrt(d2) < fm" without the square root where the answer is clear: sqrt is monotone and correctly rounded, so d2 below
  // fm^2 (1 - 2^-50) or above fm^2 (1 + 2^-50) decides; only a value inside that band takes the root itself
  const double fm2 = fm * fm, fm2_lo = fm2 * (1.0 - 0x1p-50), fm2_hi = fm2 * (1.0 + 0x1p-50);
  unsigned long long best = 0;   // (score + 1) << 32 | ~k : the largest is the first draw with the best score
  for (int k = tid; k < iterations; k += OB_T) {
    const int i1 = scr[k];
    const double px = pts[3 * i1], py = pts[3 * i1 + 1], pz = pts[3 * i1 + 2];
    int score = 0;
    for (int u = 0; u < l; u++) {
      const double dx = pts[3 * u] - px, dy = pts[3 * u + 1] - py, dz = pts[3 * u + 2] - pz;
      const double d2 = dx * dx + dy * dy + dz * dz;
      const bool in = d2 < fm2_lo ? true : (d2 > fm2_hi ? false : sqrt(d2) < fm);
      score += in ? 1 : 0;
    }
    const unsigned long long key = ((unsigned long long)(unsigned)(score + 1) << 32) | (unsigned)(0x7FFFFFFF - k);
    best = key > best ? key : best;
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)best, dd), hi = __shfl_xor((unsigned)(best >> 32), dd);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    best = o > best ? o : best;
  }
  __syncthreads();
  if ((tid & 63) == 0) sh.red64[tid >> 6] = best;
  __syncthreads();
  best = sh.red64[0];
  for (int w = 1; w < OB_T / 64; w++) best = sh.red64[w] > best ? sh.red64[w] : best;
  const int kbest = 0x7FFFFFFF - (int)(unsigned)best;
  const int i1 = scr[kbest];
  const double px = pts[3 * i1], py = pts[3 * i1 + 1], pz = pts[3 * i1 + 2];
  int cntl = 0;
  for (int u = tid; u < l; u += OB_T) {
    const double dx = pts[3 * u] - px, dy = pts[3 * u + 1] - py, dz = pts[3 * u + 2] - pz;
    const double d2 = dx * dx + dy * dy + dz * dz;
    const bool in = d2 < fm2_lo ? true : (d2 > fm2_hi ? false : sqrt(d2) < fm);
    flag_g[u] = in ? 1 : 0;
    if (in_lds) fl[u] = in ? 1 : 0;
    cntl += in ? 1 : 0;
  }
  const int ninl = ob_block_sum(cntl, sh.red);
  __syncthreads();
  if (tid == 0) {
    const uint8_t* flag = in_lds ? fl : flag_g;
    double sx = 0, sy = 0, sz = 0;
    for (int u = 0; u < l; u++) if (flag[u]) { sx += pts[3 * u]; sy += pts[3 * u + 1]; sz += pts[3 * u + 2]; }
    const double m = (double)ninl;
    mean[0] = sx / m; mean[1] = sy / m; mean[2] = sz / m;
  }
  __syncthreads();
  return ninl;
}

// ObjectState::projectOntoImageRectFromCamera (g2o_Object.cc:156-169, EnObjectCenter = 0) as cv::Rect(x, y, w, h) of truncated
// doubles; corner k on lane k & 7, extremes over the eight lanes
__device__ __forceinline__ void ob_project_box(const ObCam& C, const double* R, const double* t, const double* scale, int* box) {
  const int k = threadIdx.x & 7;
  const double b0 = (k == 0 || k == 1 || k == 4 || k == 5) ? 1.0 : -1.0;
  const double b1 = (k == 0 || k == 3 || k == 4 || k == 7) ? 1.0 : -1.0;
  const double b2 = k < 4 ? -1.0 : 1.0;
  double c[3];
  for (int r = 0; r < 3; r++) {
    const double s0 = R[3 * r] * (scale[0] * 0.5), s1 = R[3 * r + 1] * (scale[1] * 0.5), s2 = R[3 * r + 2] * (scale[2] * 0.5);
    c[r] = s0 * b0 + s1 * b1 + s2 * b2 + t[r] * 1.0;
  }
  const double fx = (double)C.fx, fy = (double)C.fy, cx = (double)C.cx, cy = (double)C.cy;
  const double p0 = fx * c[0] + 0.0 * c[1] + cx * c[2];
  const double p1 = 0.0 * c[0] + fy * c[1] + cy * c[2];
  const double p2 = 0.0 * c[0] + 0.0 * c[1] + 1.0 * c[2];
  double ulo = p0 / p2, vlo = p1 / p2, uhi = ulo, vhi = vlo;
#pragma unroll
  for (int d = 1; d <= 4; d <<= 1) {
    ulo = fmin(ulo, __shfl_xor(ulo, d)); uhi = fmax(uhi, __shfl_xor(uhi, d));
    vlo = fmin(vlo, __shfl_xor(vlo, d)); vhi = fmax(vhi, __shfl_xor(vhi, d));
  }
  box[0] = (int)ulo; box[1] = (int)vlo; box[2] = (int)(uhi - ulo); box[3] = (int)(vhi - vlo);
}

// Tracking::FineTuningUsing2dBox (Tracking.cc:1704-1786): executed by every lane of the calling wave (all lanes end with the
// same translation); t is updated in place
// One search of Tracking::FineTuningUsing2dBox: t[axis] moves by `step` towards a zero of the box metric (0: vertical centre offset,
// 1: height difference, 2: horizontal centre offset) until |metric| < 1, at most 400 times; the direction is the sign of the metric
// of the state before each move.  The moves are sequential in the reference; here the wave evaluates the next EIGHT states at once
// (8 lanes = the 8 cuboid corners of one state) on the assumption that the direction does not change, and takes the first state that
// ends the search or changes the direction - every value is produced by the same operations in the same order.  Called by one wave.
__device__ __forceinline__ int ob_box_metric(int which, const int* pb, int rcx, int rcy, int bh) {
  return which == 0 ? (pb[1] + pb[1] + pb[3]) / 2 - rcy : which == 1 ? pb[3] - bh : (pb[0] + pb[0] + pb[2]) / 2 - rcx;
}
__device__ void ob_tune_search(const ObCam& C, const double* R, const double* scale, double* t, int* pb, int axis, int which, bool plus, double step,
                               int rcx, int rcy, int bh) {
  const int g = (threadIdx.x & 63) >> 3;
  int m = ob_box_metric(which, pb, rcx, rcy, bh);
  int done = 0;
  // the predicted directions of the next eight moves: all like the first one, or - once a move has overshot the one-pixel window, where
  // the reference's search steps back and forth until its 400 moves are used up - alternating
  bool alt = false;
  while (done < 400) {
    const int dir = m < 0 ? -1 : 1;
    double tc[3] = {t[0], t[1], t[2]};
    double v = tc[axis];
    for (int s = 0; s <= g; s++) {                                        // the state after g + 1 moves
      const double inc = ((alt && (s & 1)) ? -dir : dir) * step;
      v = plus ? v + inc : v - inc;
    }
    tc[axis] = v;
    int pc[4];
    ob_project_box(C, R, tc, scale, pc);
    const int mc = ob_box_metric(which, pc, rcx, rcy, bh);
    const int predicted = (alt && ((g + 1) & 1)) ? -dir : dir;            // of the move after this state
    const bool stop = abs(mc) < 1, miss = (mc < 0 ? -1 : 1) != predicted;
    const unsigned long long ev = __ballot(stop || miss);
    int j = ev ? (int)((__ffsll((long long)ev) - 1) >> 3) : 7;            // the first state that ends the search or breaks the prediction, else all eight
    j = min(j, 400 - done - 1);
    // A search that steps back and forth returns to the same value after two moves ((t - s) + s == t, bit for bit, when nothing is
    // rounded away): the direction of a move depends on t alone, so the remaining moves repeat these two states until the 400 are
    // used up - the result is the current state or the one after one more move, by the parity of what is left.
    if (alt && !ev && j == 7) {
      const double v2 = __shfl(v, 8);                                      // the state after two moves
      if (__double_as_longlong(v2) == __double_as_longlong(t[axis])) {
        if ((400 - done) & 1) {                                            // an odd number of moves left: one more move
          t[axis] = __shfl(v, 0);
#pragma unroll
          for (int q = 0; q < 4; q++) pb[q] = __shfl(pc[q], 0);
          m = __shfl(mc, 0);
        }
        done = 400;
        break;
      }
    }
    const int src = 8 * j;
    t[axis] = __shfl(v, src);
#pragma unroll
    for (int q = 0; q < 4; q++) pb[q] = __shfl(pc[q], src);
    m = __shfl(mc, src);
    done += j + 1;
    if (abs(m) < 1) break;
    if (__shfl((int)miss, src)) alt = !alt;
  }
}
__device__ void ob_fine_tune(const ObCam& C, const ObDet& det, const double* q, const double* scale, double* t) {
  double R[9];
  se3_quat_to_R(q, R);
  const int bx = det.bbox[0], by = det.bbox[1], bw = det.bbox[2], bh = det.bbox[3];
  const int rcx = (bx + (bx + bw)) / 2, rcy = (by + (by + bh)) / 2;
  int pb[4];
  ob_project_box(C, R, t, scale, pb);
  ob_tune_search(C, R, scale, t, pb, 1, 0, false, 0.01, rcx, rcy, bh);
  if (t[2] > 8) ob_tune_search(C, R, scale, t, pb, 2, 1, true, 0.05, rcx, rcy, bh);
  ob_tune_search(C, R, scale, t, pb, 0, 2, false, 0.01, rcx, rcy, bh);
}

__device__ __forceinline__ float ob_fmax(const double* scale) {   // float fMaxDis = scale.norm()
  return (float)sqrt(scale[0] * scale[0] + scale[1] * scale[1] + scale[2] * scale[2]);
}

// The MapObjectPoints of a new ObjectKeyFrame (MapObjectInit / MapObjectReInit tail): for the inliers whose object-frame position
// lies within fmax, position (float), descriptor of the keypoint, one observation, UpdateNormalAndDepth against the keyframe's
// camera centre mPoc = -Roc * tco; the keyframe lists its points by feature index (sel[i] = 1 for the features that get a point)
__device__ void ob_keyframe_points(const ObArrays& A, const ObFrame& F, int s, int slot, size_t fo, int n, const Se3& pose, const double* pts,
                                   const int32_t* pidx, const uint8_t* flag, int l, float fmax, uint8_t* sel, float* selpo, int step, bool init, ObShared& sh) {
  const ObCam& C = A.cam;
  const int tid = threadIdx.x;
  const Se3 inv = se3_inverse(pose);
  float m[12];
  {
    double R[9];
    se3_quat_to_R(pose.q, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) m[4 * r + c] = (float)R[3 * r + c]; m[4 * r + 3] = (float)pose.t[r]; }
  }
  float poc[3];
  for (int r = 0; r < 3; r++) poc[r] = (float)(-((double)m[r] * (double)m[3] + (double)m[4 + r] * (double)m[7] + (double)m[8 + r] * (double)m[11]));
  for (int i = tid; i < n; i += OB_T) sel[i] = 0;
  __syncthreads();
  for (int u = tid; u < l; u += OB_T) {
    if (!flag[u]) continue;
    double x3do[3];
    se3_map(inv, pts + 3 * u, x3do);
    if (sqrt(x3do[0] * x3do[0] + x3do[1] * x3do[1] + x3do[2] * x3do[2]) > (double)fmax) continue;
    const int i = pidx[u];
    sel[i] = 1;
    selpo[3 * i] = (float)x3do[0]; selpo[3 * i + 1] = (float)x3do[1]; selpo[3 * i + 2] = (float)x3do[2];
  }
  __syncthreads();
  const size_t lb = ((size_t)s * A.M + slot) * A.LC;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += OB_T) {
    const int i = i0 + tid;
    const bool has = i < n && sel[i];
    int total;
    const int pid = base + ob_block_scan_excl(has ? 1 : 0, sh.red, &total);
    if (has && pid < A.LC) {
      const float po[3] = {selpo[3 * i], selpo[3 * i + 1], selpo[3 * i + 2]};
      const float v[3] = {po[0] - poc[0], po[1] - poc[1], po[2] - poc[2]};
      const double nd = sqrt((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1] + (double)v[2] * (double)v[2]);
      const float dist = (float)nd, inv_n = (float)(1.0 / nd);
      const float maxd = dist * C.sf[F.octave[fo + i]];
      for (int c = 0; c < 3; c++) { A.lm_po[3 * (lb + pid) + c] = po[c]; A.lm_normal[3 * (lb + pid) + c] = v[c] * inv_n; }
      A.lm_maxd[lb + pid] = maxd; A.lm_mind[lb + pid] = maxd / C.sf[C.nlevels - 1];
      const uint4* sd = reinterpret_cast<const uint4*>(F.desc + (fo + i) * 32);
      uint4* dd = reinterpret_cast<uint4*>(A.lm_desc + (lb + pid) * 32);
      dd[0] = sd[0]; dd[1] = sd[1];
      F.mp_valid[fo + i] = 1; F.mp_observed[fo + i] = 1; F.mp_id[fo + i] = pid; F.outlier[fo + i] = 0;
      for (int c = 0; c < 3; c++) F.mp_po[3 * (fo + i) + c] = po[c];
    }
    base += total;
  }
  __syncthreads();
  if (tid == 0) {
    ObMapObject& O = A.mobj[(size_t)s * A.M + slot];
    O.npts = base < A.LC ? base : A.LC;
    // mnLastKeyFrameId is written by MapObjectInit (Tracking.cc:1875) and CreateNewObjectKeyFrame (:2835) only: after a
    // MapObjectReInit (:1908-2031) the next frame's TrackLastFrameObjectPoint does not skip the object at :2302
    if (init) O.kf_frame = step;
    O.local_valid = 0;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Tracking::TrackMapObject for one detection (pose prediction Tcl * Tco, InitializeCurrentObjPose, FineTuningUsing2dBox, or
// MapObjectInit), then this detection's part of TrackLastFrameObjectPoint: the temporal points of its last-frame features and
// the SearchByBruceMatching problem.  One workgroup per (detection, sequence).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OB_T) void ob_track(ObArrays A, int step) {
  __shared__ ObShared sh;
  __shared__ Se3 s_pose;
  __shared__ int s_flag[4];
  __shared__ unsigned long long s_stop[OB_T / 64];
  const int s = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, K = A.K;   // sequence-major launch: see psk_ob_track
  const ObCam& C = A.cam;
  ObFrame& F = A.cur;
  BfProb* bp = A.bf_prob + (size_t)s * K + j;
  if (tid == 0) *bp = BfProb{0, 0, 0, 0};
  if (j >= F.ndet[s] || !ob_camera_initialized(A, s, step)) return;
  const size_t fb = (size_t)s * A.OC;
  const int b0 = F.off[(size_t)s * (K + 1) + j], n = F.off[(size_t)s * (K + 1) + j + 1] - b0;
  const size_t fo = fb + b0;
  const ObDet det = F.det[(size_t)s * K + j];
  const int moslot = F.mo[(size_t)s * K + j];
  if (moslot == -1) return;                                         // the MapObject table is full: the detection is ignored
  double* pts = A.cam_pts + 3 * fo;
  int32_t* pidx = A.inl_flag + fo;                                  // feature index of point u
  int32_t* scr = A.pj_match + fo;                                   // scratch of this detection's range: draws / per-feature flags
  uint8_t* flag = reinterpret_cast<uint8_t*>(A.bf_qot + fo);        // inlier flags (bytes) in the range's bf_qot slots
  ObStat* st = A.stats + ((size_t)step * A.S + s) * K + j;
  int l;
  ob_cam_points(A, F, fo, n, pts, pidx, &l, sh);
  const float fmax_det = ob_fmax(det.scale);
  if (moslot < 0) {
    // ---- Tracking::MapObjectInit (Tracking.cc:1787-1930) ----
    const int slot = -2 - moslot;
    const int ninl = ob_ransac(pts, l, fmax_det, (int)(0.8 * l), scr, flag, sh.mean, sh);
    bool ok = ninl >= 3;
    double c[3] = {sh.mean[0], sh.mean[1], sh.mean[2]};
    if (ok && c[2] < 8) ok = false;
    if (!ok) { if (tid == 0) F.mo[(size_t)s * K + j] = -1; return; }
    c[2] += 0.2 * det.scale[0];
    c[1] = 0 + det.scale[1] / 2;
    const Se3 truth = ob_pose(det.pose7);
    if (tid < 64) {
      double t[3] = {c[0], c[1], c[2]};
      ob_fine_tune(C, det, truth.q, det.scale, t);
      if (tid == 0) {
        Se3 P = truth;
        P.t[0] = t[0]; P.t[1] = t[1]; P.t[2] = t[2];
        s_pose = P;
        ObMapObject& O = A.mobj[(size_t)s * A.M + slot];
        O.id = det.id; O.first_frame = step; O.tco_frame = step;
        O.dyn = 3;     // new MapObject(id, candidate_cuboid->GetDynamicFlag() = true, ...): mbFirstObserved, no history (MapObject.cc:20-21)
        for (int k = 0; k < 3; k++) O.scale[k] = det.scale[k];
        ob_store_pose(O.tco, P);
        F.mo[(size_t)s * K + j] = slot;
        st->tracked = 1; st->is_new = 1;
      }
    }
    __syncthreads();
    const Se3 pose = s_pose;
    ob_keyframe_points(A, F, s, slot, fo, n, pose, pts, pidx, flag, l, fmax_det, A.occupied + fo, A.po_obs + 3 * fo, step, true, sh);
    return;
  }
  // ---- an object seen before (Tracking.cc:1553-1622) ----
  ObMapObject& O = A.mobj[(size_t)s * A.M + moslot];
  if (tid == 0) {
    // camera_Tcl = mCurrentFrame.mTcw * mLastFrame.mTwc when both frames have a pose, else the identity
    const float* Fc = ob_cam_pose(A, s, step);
    const float* Lc = ob_cam_pose(A, s, step - 1);
    Se3 tcl;
    tcl.q[0] = tcl.q[1] = tcl.q[2] = 0; tcl.q[3] = 1; tcl.t[0] = tcl.t[1] = tcl.t[2] = 0;
    if (Fc && Lc) {
      float twc[16], M4[16];
      for (int i = 0; i < 16; i++) twc[i] = (i % 5 == 0) ? 1.f : 0.f;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) twc[4 * r + c] = Lc[4 * c + r];
      for (int r = 0; r < 3; r++) {
        float acc = 0;
        for (int c = 0; c < 3; c++) acc += Lc[4 * c + r] * Lc[4 * c + 3];
        twc[4 * r + 3] = -acc;
      }
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
          float acc = 0;
          for (int k = 0; k < 4; k++) acc += Fc[4 * r + k] * twc[4 * k + c];
          M4[4 * r + c] = acc;
        }
      tcl = se3_from_mat4f(M4);
    }
    s_pose = se3_mul(tcl, ob_pose(O.tco));
  }
  __syncthreads();
  {
    // InitializeCurrentObjPose (Tracking.cc:1640-1702): the RANSAC centroid of the detection's stereo points replaces the translation
    const int ninl = ob_ransac(pts, l, fmax_det, (int)(0.8 * l), scr, flag, sh.mean, sh);
    Se3 pose = s_pose;
    if (ninl >= 3) {
      double c[3] = {sh.mean[0], sh.mean[1], sh.mean[2]};
      if (c[2] > 8) c[2] += 0.2 * det.scale[0];
      c[1] = 0 + det.scale[1] / 2;
      pose.t[0] = c[0]; pose.t[1] = c[1]; pose.t[2] = c[2];
    }
    __syncthreads();
    if (tid < 64) {
      double t[3] = {pose.t[0], pose.t[1], pose.t[2]};
      ob_fine_tune(C, det, pose.q, O.scale, t);
      if (tid == 0) {
        pose.t[0] = t[0]; pose.t[1] = t[1]; pose.t[2] = t[2];
        s_pose = pose;
        const int latest = O.tco_frame;
        ob_store_pose(O.tco, pose);
        O.tco_frame = step;
        A.tracked[(size_t)s * K + j] = 1;
        st->tracked = 1;
        int lj = -1;
        if (latest == step - 1) {
          const int nl = A.last.ndet[s];
          for (int k = 0; k < nl; k++) if (A.last.det[(size_t)s * K + k].id == det.id) { lj = k; break; }
        }
        A.in_last[(size_t)s * K + j] = lj;
        if (lj >= 0) for (int c = 0; c < 7; c++) A.last_tco[((size_t)s * K + j) * 7 + c] = A.last.tco[((size_t)s * K + lj) * 7 + c];
        s_flag[0] = lj;
      }
    }
    __syncthreads();
  }
  const int lj = s_flag[0];
  if (lj < 0) return;
  // ---- TrackLastFrameObjectPoint, first part (Tracking.cc:2296-2366): temporal MapObjectPoints of the last frame's features ----
  ObFrame& L = A.last;
  const int lb0 = L.off[(size_t)s * (K + 1) + lj], ln = L.off[(size_t)s * (K + 1) + lj + 1] - lb0;
  const size_t lo = fb + lb0;
  if (!(O.kf_frame == step - 1 || O.first_frame == step - 1)) {
    const Se3 inv = se3_inverse(ob_pose(L.tco + ((size_t)s * K + lj) * 7));
    const float fmax_mo = ob_fmax(O.scale);
    const float thr = 2 * C.th_depth;
    // the reference walks the features by (depth, index) and stops behind the first one beyond 2 * mThDepth that it did not skip
    // (a feature whose new point would lie farther than fmax from the object centre is skipped with `continue`, past the stop test)
    unsigned long long stop = ~0ull;
    for (int i = tid; i < ln; i += OB_T) {
      const float d = L.depth[lo + i];
      if (!(d > 0)) continue;
      bool skipped = false;
      if (!L.mp_valid[lo + i] || !L.mp_observed[lo + i]) {
        const float xx = (L.x[lo + i] - C.cx) * d * C.inv_fx, yy = (L.y[lo + i] - C.cy) * d * C.inv_fy;
        const double pc[3] = {(double)xx, (double)yy, (double)d};
        double po[3];
        se3_map(inv, pc, po);
        const float pf[3] = {(float)po[0], (float)po[1], (float)po[2]};
        const float nf = sqrtf(pf[0] * pf[0] + pf[1] * pf[1] + pf[2] * pf[2]);
        skipped = nf > fmax_mo;
      }
      if (!skipped && d > thr) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
        stop = key < stop ? key : stop;
      }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
      const unsigned lo32 = __shfl_xor((unsigned)stop, dd), hi32 = __shfl_xor((unsigned)(stop >> 32), dd);
      const unsigned long long o = ((unsigned long long)hi32 << 32) | lo32;
      stop = o < stop ? o : stop;
    }
    if ((tid & 63) == 0) s_stop[tid >> 6] = stop;
    __syncthreads();
    stop = s_stop[0];
    for (int w = 1; w < OB_T / 64; w++) stop = s_stop[w] < stop ? s_stop[w] : stop;
    for (int i = tid; i < ln; i += OB_T) {
      const float d = L.depth[lo + i];
      if (!(d > 0)) continue;
      const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
      if (key > stop) continue;
      if (!L.mp_valid[lo + i] || !L.mp_observed[lo + i]) {
        const float xx = (L.x[lo + i] - C.cx) * d * C.inv_fx, yy = (L.y[lo + i] - C.cy) * d * C.inv_fy;
        const double pc[3] = {(double)xx, (double)yy, (double)d};
        double po[3];
        se3_map(inv, pc, po);
        const float pf[3] = {(float)po[0], (float)po[1], (float)po[2]};
        const float nf = sqrtf(pf[0] * pf[0] + pf[1] * pf[1] + pf[2] * pf[2]);
        if (nf > fmax_mo) continue;
        L.mp_valid[lo + i] = 1; L.mp_observed[lo + i] = 0; L.mp_id[lo + i] = -1;
        for (int c = 0; c < 3; c++) L.mp_po[3 * (lo + i) + c] = pf[c];
      }
    }
    __syncthreads();
  }
  // ---- the SearchByBruceMatching problem (ORBmatcher.cc:2043-2155): queries = the last frame's features of the object ----
  for (int i = tid; i < ln; i += OB_T) A.bf_qvalid[lo + i] = (L.mp_valid[lo + i] && !L.outlier[lo + i]) ? 1 : 0;
  if (tid == 0) *bp = BfProb{(int32_t)lo, ln, (int32_t)fo, n};
}

// ---------------------------------------------------------------------------------------------------------------------
// After the brute-force matcher: the matches become the frame's MapObjectPoints (Tracking.cc:2388-2390), and the first
// CFSE3ObjStateOptimization problem over the detections with at least 10 matches (:2391-2425).  One workgroup per sequence.
// ---------------------------------------------------------------------------------------------------------------------
__device__ void ob_fill_cfse3(const ObArrays& A, int s, const int32_t* which, int step) {
  const int tid = threadIdx.x, K = A.K;
  const ObFrame& F = A.cur;
  const ObCam& C = A.cam;
  const size_t fb = (size_t)s * A.OC;
  const int ntot = F.off[(size_t)s * (K + 1) + K];
  for (int i = tid; i < ntot; i += OB_T) {
    A.po_obs[3 * (fb + i)] = F.x[fb + i]; A.po_obs[3 * (fb + i) + 1] = F.y[fb + i]; A.po_obs[3 * (fb + i) + 2] = F.uright[fb + i];
    A.po_is2[fb + i] = C.inv_sigma2[F.octave[fb + i]];
  }
  if (tid == 0) {
    int nv = 0;
    const int nd = F.ndet[s];
    for (int j = 0; j < nd; j++) {
      if (!which[(size_t)s * K + j]) continue;
      const int slot = F.mo[(size_t)s * K + j];
      const int b0 = F.off[(size_t)s * (K + 1) + j], b1 = F.off[(size_t)s * (K + 1) + j + 1];
      A.po_vert[(size_t)s * K + nv] = PoVertex{(int32_t)(fb + b0), (int32_t)(fb + b1)};
      const double* p = A.mobj[(size_t)s * A.M + slot].tco;
      for (int c = 0; c < 7; c++) A.po_pose[((size_t)s * K + nv) * 7 + c] = p[c];
      A.po_vmap[(size_t)s * K + nv] = j;
      nv++;
    }
    A.po_prob[s] = PoProb{(int32_t)(s * K), nv, 1, C.fx, C.fy, C.cx, C.cy, C.mbf};
    A.po_result[s] = 0;
  }
}

__global__ __launch_bounds__(OB_T) void ob_after_bf(ObArrays A, int step) {
  const int s = blockIdx.x, tid = threadIdx.x, K = A.K;
  ObFrame& F = A.cur;
  const ObFrame& L = A.last;
  const size_t fb = (size_t)s * A.OC;
  const int nd = F.ndet[s];
  for (int j = 0; j < nd; j++) {
    const int lj = A.in_last[(size_t)s * K + j];
    if (lj < 0) continue;
    const int b0 = F.off[(size_t)s * (K + 1) + j], n = F.off[(size_t)s * (K + 1) + j + 1] - b0;
    const size_t lo = fb + L.off[(size_t)s * (K + 1) + lj];
    for (int t = tid; t < n; t += OB_T) {
      const int q = A.bf_qot[fb + b0 + t];
      const size_t o = fb + b0 + t;
      F.mp_valid[o] = q >= 0; F.mp_observed[o] = 0; F.mp_id[o] = -1;
      if (q >= 0) {
        F.mp_observed[o] = L.mp_observed[lo + q]; F.mp_id[o] = L.mp_id[lo + q];
        for (int c = 0; c < 3; c++) F.mp_po[3 * o + c] = L.mp_po[3 * (lo + q) + c];
      }
    }
    if (tid == 0) {
      const int nm = A.bf_nmatch[(size_t)s * K + j];
      A.stats[((size_t)step * A.S + s) * K + j].bf_matches = nm;
      A.need[(size_t)s * K + j] = nm >= 10 ? 1 : 0;
    }
  }
  __syncthreads();
  ob_fill_cfse3(A, s, A.need, step);
}

// ---------------------------------------------------------------------------------------------------------------------
// After the first CFSE3: the optimised poses, the outliers discarded (Tracking.cc:2426-2462), then TrackObjectLocalMap's search:
// UpdateObjectLocalKeyFrames / Points (the keyframe of the object's (re-)initialisation), Frame::isInFrustum(pMP, nOrder, 0.5)
// (Frame.cc:1744-1790) and the SearchByProjection(F, nOrder, MOPs, th = 1) problem (Tracking.cc:2522-2575).
// One workgroup per (detection, sequence).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OB_T) void ob_after_cf1(ObArrays A, int step) {
  __shared__ int red[OB_T / 64];
  extern __shared__ uint8_t seen[];
  const int s = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, K = A.K;   // sequence-major launch: see psk_ob_track
  const ObCam& C = A.cam;
  ObFrame& F = A.cur;
  PjProb* pp = A.pj_prob + (size_t)s * K + j;
  if (tid == 0) {
    PjProb d;
    for (int i = 0; i < (int)(sizeof(PjProb) / 4); i++) ((int32_t*)&d)[i] = 0;
    *pp = d;
  }
  if (j >= F.ndet[s] || !A.tracked[(size_t)s * K + j]) return;
  const size_t fb = (size_t)s * A.OC;
  const int b0 = F.off[(size_t)s * (K + 1) + j], n = F.off[(size_t)s * (K + 1) + j + 1] - b0;
  const size_t fo = fb + b0;
  const int slot = F.mo[(size_t)s * K + j];
  ObMapObject& O = A.mobj[(size_t)s * A.M + slot];
  ObStat* st = A.stats + ((size_t)step * A.S + s) * K + j;
  if (A.need[(size_t)s * K + j]) {
    if (A.po_result[s] && tid == 0) {
      const int nv = A.po_prob[s].k;
      for (int v = 0; v < nv; v++)
        if (A.po_vmap[(size_t)s * K + v] == j) for (int c = 0; c < 7; c++) O.tco[c] = A.po_pose[((size_t)s * K + v) * 7 + c];
    }
    int nmap = 0;
    for (int i = tid; i < n; i += OB_T) {
      if (!F.mp_valid[fo + i]) continue;
      if (F.outlier[fo + i]) { F.mp_valid[fo + i] = 0; F.outlier[fo + i] = 0; continue; }
      if (F.mp_observed[fo + i]) nmap++;
    }
    nmap = ob_block_sum(nmap, red);
    if (tid == 0) A.track_ok[(size_t)s * K + j] = nmap >= 10 ? 1 : 0;
  }
  __syncthreads();
  // ---- SearchObjectLocalPoints ----
  int anyobs = 0;
  for (int i = tid; i < n; i += OB_T) anyobs |= (F.mp_valid[fo + i] && F.mp_observed[fo + i]) ? 1 : 0;
  anyobs = ob_block_sum(anyobs, red);
  if (anyobs && tid == 0) O.local_valid = 1;
  __syncthreads();
  const int nloc = O.local_valid ? O.npts : 0;
  for (int i = tid; i < nloc; i += OB_T) seen[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += OB_T) {
    const int id = F.mp_id[fo + i];
    if (id >= 0 && id < nloc) seen[id] = 1;        // mnLastFrameSeen: the frame's points and the outliers just discarded
  }
  __syncthreads();
  const ObDet det = F.det[(size_t)s * K + j];
  const double bx0 = (double)det.bbox[0], by0 = (double)det.bbox[1];
  const double bx1 = (double)det.bbox[0] + (double)det.bbox[2], by1 = (double)det.bbox[1] + (double)det.bbox[3];
  const Se3 tco = ob_pose(O.tco);
  const Se3 inv = se3_inverse(tco);
  const float poc[3] = {(float)inv.t[0], (float)inv.t[1], (float)inv.t[2]};
  const size_t lb = ((size_t)s * A.M + slot) * A.LC;
  int nto = 0;
  for (int i = tid; i < nloc; i += OB_T) {
    const size_t q = lb + i;
    A.pj_qvalid[q] = 0; A.pj_qu[q] = 0.f; A.pj_qv[q] = 0.f; A.pj_qur[q] = 0.f; A.pj_qrad[q] = 5.f; A.pj_qrer[q] = 0.f; A.pj_qminl[q] = 0; A.pj_qmaxl[q] = 0;
    if (seen[i]) continue;
    const float* po = A.lm_po + 3 * q;
    const double pod[3] = {(double)po[0], (double)po[1], (double)po[2]};
    double pc[3];
    se3_map(tco, pod, pc);
    const float X = (float)pc[0], Y = (float)pc[1], Z = (float)pc[2];
    if (Z < 0) continue;
    const float invz = 1.0f / Z;
    const float u = C.fx * X * invz + C.cx, v = C.fy * Y * invz + C.cy;
    if (!((double)u >= bx0 && (double)u < bx1 && (double)v >= by0 && (double)v < by1)) continue;
    const float d[3] = {po[0] - poc[0], po[1] - poc[1], po[2] - poc[2]};
    const float dist = (float)sqrt((double)d[0] * (double)d[0] + (double)d[1] * (double)d[1] + (double)d[2] * (double)d[2]);
    const float maxd = 1.2f * A.lm_maxd[q], mind = 0.8f * A.lm_mind[q];
    if (dist < mind || dist > maxd) continue;
    const float* pn = A.lm_normal + 3 * q;
    const double dot = (double)d[0] * (double)pn[0] + (double)d[1] * (double)pn[1] + (double)d[2] * (double)pn[2];
    const float viewCos = (float)(dot / (double)dist);
    if (viewCos < 0.5f) continue;
    const float ratio = A.lm_maxd[q] / dist;
    int level = (int)ceilf((float)log((double)ratio) / C.log_sf);
    level = level < 0 ? 0 : (level >= C.nlevels ? C.nlevels - 1 : level);
    const float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;
    A.pj_qvalid[q] = 1; A.pj_qu[q] = u; A.pj_qv[q] = v; A.pj_qur[q] = u - C.mbf * invz;
    A.pj_qrer[q] = r * C.sf[level]; A.pj_qminl[q] = level - 1; A.pj_qmaxl[q] = level + 1;
    nto++;
  }
  nto = ob_block_sum(nto, red);
  if (tid == 0) st->lm_candidates = nto;
  if (nto > 0) {
    for (int i = tid; i < n; i += OB_T) {
      const double x = (double)F.x[fo + i], y = (double)F.y[fo + i];
      A.inbbox[fo + i] = (x >= bx0 && x < bx1 && y >= by0 && y < by1) ? 1 : 0;
      A.occupied[fo + i] = (F.mp_valid[fo + i] && F.mp_observed[fo + i]) ? 1 : 0;
    }
    if (tid == 0) {
      PjProb d;
      for (int i = 0; i < (int)(sizeof(PjProb) / 4); i++) ((int32_t*)&d)[i] = 0;
      d.t_off = (int32_t)fo; d.nt = n; d.q_off = (int32_t)lb; d.nq = nloc; d.c_off = (int32_t)(((size_t)s * K + j) * A.LC);
      d.grid_off = (int32_t)(((size_t)s * K + j) * (OB_NCELL + 1));
      d.min_x = 0.f; d.min_y = 0.f; d.gw_inv = C.gw_inv; d.gh_inv = C.gh_inv;
      d.th_dist = 130; d.ratio_test = 1; d.nn_ratio = 0.8f; d.check_ori = 0; d.use_bbox = 1; d.frame_mode = 0;
      d.fx = C.fx; d.fy = C.fy; d.cx = C.cx; d.cy = C.cy; d.mbf = C.mbf; d.mb = C.mb;
      for (int l = 0; l < 8; l++) d.scale[l] = l < C.nlevels ? C.sf[l] : 1.f;
      d.th = 1.f;
      *pp = d;
    }
  }
}

// After the windowed matcher: the local-map matches join the frame's points (ORBmatcher.cc:236-239), then the second CFSE3 problem
// over every tracked detection (Tracking.cc:2473-2491).  One workgroup per sequence.
__global__ __launch_bounds__(OB_T) void ob_after_lm(ObArrays A, int step) {
  const int s = blockIdx.x, tid = threadIdx.x, K = A.K;
  ObFrame& F = A.cur;
  const size_t fb = (size_t)s * A.OC;
  const int nd = F.ndet[s];
  for (int j = 0; j < nd; j++) {
    if (!A.tracked[(size_t)s * K + j] || A.pj_prob[(size_t)s * K + j].nq == 0) continue;
    const int b0 = F.off[(size_t)s * (K + 1) + j], n = F.off[(size_t)s * (K + 1) + j + 1] - b0;
    const int slot = F.mo[(size_t)s * K + j];
    const size_t lb = ((size_t)s * A.M + slot) * A.LC;
    for (int t = tid; t < n; t += OB_T) {
      const int m = A.pj_match[fb + b0 + t];
      if (m < 0) continue;
      const size_t o = fb + b0 + t;
      F.mp_valid[o] = 1; F.mp_observed[o] = 1; F.mp_id[o] = m;
      for (int c = 0; c < 3; c++) F.mp_po[3 * o + c] = A.lm_po[3 * (lb + m) + c];
    }
    if (tid == 0) A.stats[((size_t)step * A.S + s) * K + j].lm_matches = A.pj_nmatch[(size_t)s * K + j];
  }
  __syncthreads();
  ob_fill_cfse3(A, s, A.tracked, step);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tracking::DynamicStaticDiscrimination (Tracking.cc:2058-2202; Track calls it right behind the object tracking functions, :1244) for
// one detection that was tracked from the last frame; its tail StaticPointRecoveryFromObj is outside the slice.
//   gates: camera-frame depth of the object outside [7, mThDepth] -> the detection takes its MapObject's flag; the box within
//   width / 2 + 60 px of the image centre column -> dynamic by prior (a vehicle straight ahead).
//   test: every MapObjectPoint of the detection moved as if the object stood still (Pc = Tcw_cur Tcw_last^-1 Tco_last Po) against its
//   observation - the arithmetic of dyn_kernels.hip (a f-4), term by term; per kind (monocular / stereo) the values are sorted, those
//   above 5 x median dropped and the rest summed IN SORTED ORDER (std::sort + std::accumulate), here by ranking: a value's rank among
//   its kind is its sorted position (equal values are interchangeable in a sum).
//   flags: DetectionObject::SetDynamicFlag(mono, stereo) (DetectionObject.cc:169-196), MapObject::DynamicDetection / SetDynamicFlag
//   (MapObject.cc:414-448) with its queue of the last four verdicts.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ob_mo_dynamic_detection(int d, bool f) {
  if (d & 4) return d;
  int len = (d >> 8) & 7, h = (d >> 4) & 15;
  if (len < 4) { h |= (f ? 1 : 0) << len; len++; }
  else h = (h >> 1) | ((f ? 1 : 0) << 3);                       // push, then pop the oldest
  d = (d & ~0x7F0) | (h << 4) | (len << 8);
  if (len < 4) return d;
  if (h == (f ? 15 : 0) && (d & 1) != (f ? 1 : 0)) d |= 4;      // four equal verdicts against the current flag: mbDynamicChanged
  return d;
}
__device__ __forceinline__ int ob_mo_set_dynamic(int d, bool f) {
  if (d & 2) d = (d & ~3) | (f ? 1 : 0);
  if (d & 4) d = (d & ~5) | (f ? 1 : 0);
  return d;
}

__device__ void ob_dsd(const ObArrays& A, const ObFrame& F, const ObFrame& L, int s, int j, int lj, ObMapObject& O, size_t fo, int n, int step,
                       ObStat* st, ObShared& sh, double* s_avg, int* s_kept) {
  const ObCam& C = A.cam;
  const int tid = threadIdx.x, K = A.K;
  const ObDet det = F.det[(size_t)s * K + j];
  const int mo_dyn = O.dyn;
  const double depth = O.tco[2];
  if (depth < 7 || depth > (double)C.th_depth) {
    if (tid == 0) st->dynamic = mo_dyn & 1;
    return;
  }
  const double middle_x = (double)(C.w / 2), current_px = (double)(det.bbox[0] + det.bbox[2] / 2);
  if (fabs(current_px - middle_x) < (double)(det.bbox[2] / 2 + 60)) {
    if (tid == 0) { st->dynamic = 1; O.dyn = ob_mo_set_dynamic(mo_dyn, true); }
    return;
  }
  // current_pose * last_pose.inverse() on the frames' mSETcw (a frame without a pose keeps the identity)
  Se3 ident;
  ident.q[0] = ident.q[1] = ident.q[2] = 0; ident.q[3] = 1; ident.t[0] = ident.t[1] = ident.t[2] = 0;
  const float* Fc = ob_cam_pose(A, s, step);
  const float* Lc = ob_cam_pose(A, s, step - 1);
  const Se3 Tc = Fc ? se3_from_mat4f(Fc) : ident, Tl = Lc ? se3_from_mat4f(Lc) : ident;
  Se3 Tli;
  Tli.q[0] = -Tl.q[0]; Tli.q[1] = -Tl.q[1]; Tli.q[2] = -Tl.q[2]; Tli.q[3] = Tl.q[3];
  const double nt[3] = {Tl.t[0] * -1., Tl.t[1] * -1., Tl.t[2] * -1.};
  se3_rotate(Tli.q, nt, Tli.t);
  const Se3 trel = se3_mul(Tc, Tli);
  const Se3 tco = ob_pose(A.last_tco + ((size_t)s * K + j) * 7);   // ob_track's copy: block (s, lj) of this launch may already have replaced L.tco[lj]
  double* vals = A.cam_pts + 3 * fo;                                    // [n] values, [n] monocular sorted, [n] stereo sorted
  uint8_t* kind = reinterpret_cast<uint8_t*>(A.bf_qot + fo);           // 0 no point, 1 monocular, 2 stereo
  const double fx = (double)C.fx, fy = (double)C.fy, cx = (double)C.cx, cy = (double)C.cy;
  int cm = 0, cs = 0;
  for (int i = tid; i < n; i += OB_T) {
    uint8_t kd = 0;
    double v = 0;
    if (F.mp_valid[fo + i]) {
      const double po[3] = {(double)F.mp_po[3 * (fo + i)], (double)F.mp_po[3 * (fo + i) + 1], (double)F.mp_po[3 * (fo + i) + 2]};
      double Plc[3], Pc[3];
      se3_map(tco, po, Plc);
      se3_map(trel, Plc, Pc);
      const double invz = 1.0 / Pc[2];
      const double w = (double)C.inv_sigma2[F.octave[fo + i]];
      const double z0 = cx + Pc[0] * invz * fx, z1 = cy + Pc[1] * invz * fy;
      const double e0 = (double)F.x[fo + i] - z0, e1 = (double)F.y[fo + i] - z1;
      const float ur = F.uright[fo + i];
      if (ur < 0) { v = e0 * (w * e0) + e1 * (w * e1); kd = 1; cm++; }
      else {
        const double z2 = z0 - (double)C.mbf * invz;
        const double e2 = (double)ur - z2;
        v = e0 * (w * e0) + e1 * (w * e1) + e2 * (w * e2); kd = 2; cs++;
      }
    }
    vals[i] = v; kind[i] = kd;
  }
  cm = ob_block_sum(cm, sh.red);
  cs = ob_block_sum(cs, sh.red);
  __syncthreads();
  for (int i = tid; i < n; i += OB_T) {
    const int kd = kind[i];
    if (!kd) continue;
    const double v = vals[i];
    int r = 0;
    for (int k = 0; k < n; k++) {
      const double vk = vals[k];
      r += (kind[k] == kd && (vk < v || (vk == v && k < i))) ? 1 : 0;
    }
    vals[(size_t)kd * n + r] = v;
  }
  __syncthreads();
  if (tid < 128) {
    // wave 0: the monocular list, wave 1: the stereo one.  The sum runs in sorted order as std::accumulate does; the values come 64 at a
    // time into the lanes' registers and are added one by one out of them (v_readlane: a dependent global load per element would cost
    // a cache round trip each - 90 us for 300 points).  A sorted list keeps a PREFIX (what is above 5 x median sits at the end).
    const int which = tid >> 6, lane = tid & 63, num = which ? cs : cm;
    const double* v = vals + (size_t)(which + 1) * n;
    double avg = 0;
    int kept = num;
    if (num >= 5) {
      const double cut = 5 * v[num / 2];                       // int(size / 2 + 0.5) with integer size / 2
      kept = 0;
      double sum = 0.0;
      for (int base = 0; base < num; base += 64) {
        const bool in = base + lane < num;
        const double x = in ? v[base + lane] : 0.0;
        const int c = __popcll(__ballot(in && !(x > cut)));
        const int xlo = __double2loint(x), xhi = __double2hiint(x);
        for (int i = 0; i < c; i++) sum += __hiloint2double(__builtin_amdgcn_readlane(xhi, i), __builtin_amdgcn_readlane(xlo, i));
        kept += c;
        if (c < 64) break;
      }
      avg = sum / kept;
    }
    if (lane == 0) { s_avg[which] = avg; s_kept[which] = kept; }
  }
  __syncthreads();
  if (tid == 0) {
    const double mono = s_avg[0], stereo = s_avg[1];
    st->dyn_n_mono = s_kept[0]; st->dyn_n_stereo = s_kept[1];
    if (mono > 0 || stereo > 0) {
      const bool f = mono > 1 || stereo > 2;
      st->dyn_mono = mono; st->dyn_stereo = stereo; st->dynamic = f ? 1 : 0;
      O.dyn = ob_mo_set_dynamic(ob_mo_dynamic_detection(mo_dyn, f), f);
    } else {
      st->dynamic = mo_dyn & 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The end of the object chain: the second CFSE3's poses and inliers (Tracking.cc:2492-2520), the end of Track for SLOT mode 4
// (:1443-1478: matches on temporal points dropped, MapObjectReInit for a detection whose tracking failed, :1932-2031), the
// frame's statistics, and mLastFrame = Frame(mCurrentFrame).  One workgroup per (detection, sequence).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OB_T) void ob_finish(ObArrays A, int step) {
  __shared__ ObShared sh;
  __shared__ Se3 s_pose;
  __shared__ int s_int[2];
  __shared__ double s_dsd[2];
  const int s = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, K = A.K;   // sequence-major launch: see psk_ob_track
  const ObCam& C = A.cam;
  ObFrame& F = A.cur;
  ObFrame& L = A.last;
  const size_t fb = (size_t)s * A.OC;
  const int nd = F.ndet[s];
  if (tid == 0) {   // windows of SearchByProjection(F, nOrder, MOPs) that held more candidates than the store: counted, then re-armed
    const int ov = A.pj_overflow[(size_t)s * K + j];
    if (ov) { atomicAdd(&A.search_overflow[s], ov); A.pj_overflow[(size_t)s * K + j] = 0; }
  }
  if (j < nd) {
    const int b0 = F.off[(size_t)s * (K + 1) + j], n = F.off[(size_t)s * (K + 1) + j + 1] - b0;
    const size_t fo = fb + b0;
    const int slot = F.mo[(size_t)s * K + j];
    ObStat* st = A.stats + ((size_t)step * A.S + s) * K + j;
    if (slot >= 0) {
      ObMapObject& O = A.mobj[(size_t)s * A.M + slot];
      if (A.tracked[(size_t)s * K + j]) {
        if (A.po_result[s] && tid == 0) {
          const int nv = A.po_prob[s].k;
          for (int v = 0; v < nv; v++)
            if (A.po_vmap[(size_t)s * K + v] == j) for (int c = 0; c < 7; c++) O.tco[c] = A.po_pose[((size_t)s * K + v) * 7 + c];
        }
        int inl = 0;
        for (int i = tid; i < n; i += OB_T) {
          if (!F.mp_valid[fo + i]) continue;
          if (F.outlier[fo + i]) F.mp_valid[fo + i] = 0;
          else if (F.mp_observed[fo + i]) inl++;
        }
        inl = ob_block_sum(inl, sh.red);
        if (tid == 0) { st->inliers = inl; A.track_ok[(size_t)s * K + j] = inl > 10 ? 1 : 0; }
      }
      __syncthreads();
      {
        // Moving Objects Recognition (Tracking.cc:1244): the detections tracked from the last frame; one tracked from an older frame took
        // its MapObject's flag in TrackMapObject (:1617)
        const int lj = A.in_last[(size_t)s * K + j];
        if (lj >= 0) ob_dsd(A, F, L, s, j, lj, O, fo, n, step, st, sh, s_dsd, s_int);
        else if (A.tracked[(size_t)s * K + j] && tid == 0) st->dynamic = O.dyn & 1;
        __syncthreads();
      }
      if (O.first_frame != step) {
        for (int i = tid; i < n; i += OB_T)
          if (F.mp_valid[fo + i] && !F.mp_observed[fo + i]) { F.mp_valid[fo + i] = 0; F.outlier[fo + i] = 0; }
        __syncthreads();
        if (!A.track_ok[(size_t)s * K + j]) {
          // ---- Tracking::MapObjectReInit ----
          if (tid == 0) { st->reinit = 1; O.npts = 0; }
          for (int i = tid; i < n; i += OB_T) { F.mp_valid[fo + i] = 0; F.mp_observed[fo + i] = 0; F.mp_id[fo + i] = -1; F.outlier[fo + i] = 0; }
          const ObDet det = F.det[(size_t)s * K + j];
          const float fmax_det = ob_fmax(det.scale);
          double* pts = A.cam_pts + 3 * fo;
          int32_t* pidx = A.inl_flag + fo;
          int32_t* scr = A.pj_match + fo;
          uint8_t* flag = reinterpret_cast<uint8_t*>(A.bf_qot + fo);
          // the features with depth in (depth, index) order up to the first one beyond 2 * mThDepth past the 100th
          const float thr = 2 * C.th_depth;
          int l = 0;
          {
            // rank of every feature among those with depth; the list ends at the smallest rank r with depth > thr and r + 1 > 100
            int* rank = scr;
            for (int i = tid; i < n; i += OB_T) {
              const float d = F.depth[fo + i];
              int r = -1;
              if (d > 0) {
                r = 0;
                for (int k = 0; k < n; k++) { const float dk = F.depth[fo + k]; r += (dk > 0 && (dk < d || (dk == d && k < i))) ? 1 : 0; }
              }
              rank[i] = r;
            }
            __syncthreads();
            int stop = 1 << 30, cntd = 0;
            for (int i = tid; i < n; i += OB_T) {
              const int r = rank[i];
              if (r < 0) continue;
              cntd++;
              if (F.depth[fo + i] > thr && r + 1 > 100) stop = min(stop, r);
            }
            cntd = ob_block_sum(cntd, sh.red);
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) stop = min(stop, __shfl_xor(stop, dd));
            __syncthreads();
            if ((tid & 63) == 0) sh.red[tid >> 6] = stop;
            __syncthreads();
            stop = sh.red[0];
            for (int w = 1; w < OB_T / 64; w++) stop = min(stop, sh.red[w]);
            l = stop < cntd ? stop + 1 : cntd;
            __syncthreads();
            for (int i = tid; i < n; i += OB_T) {
              const int r = rank[i];
              if (r < 0 || r >= l) continue;
              const float z = F.depth[fo + i];
              const float xx = (F.x[fo + i] - C.cx) * z * C.inv_fx, yy = (F.y[fo + i] - C.cy) * z * C.inv_fy;
              pts[3 * r] = (double)xx; pts[3 * r + 1] = (double)yy; pts[3 * r + 2] = (double)z;
              pidx[r] = i;
            }
            __syncthreads();
          }
          const int ninl = ob_ransac(pts, l, fmax_det, l, scr, flag, sh.mean, sh);
          bool ok = ninl > 3;
          double c[3] = {sh.mean[0], sh.mean[1], sh.mean[2]};
          if (ok && c[2] < 8) ok = false;
          if (ok) {
            if (c[2] > 8) c[2] += 0.2 * det.scale[0];
            c[1] = 0 + det.scale[1] / 2;
            const Se3 truth = ob_pose(det.pose7);
            if (tid < 64) {
              double t[3] = {c[0], c[1], c[2]};
              ob_fine_tune(C, det, truth.q, det.scale, t);
              if (tid == 0) {
                Se3 P = truth;
                P.t[0] = t[0]; P.t[1] = t[1]; P.t[2] = t[2];
                s_pose = P;
                ob_store_pose(O.tco, P);
                O.tco_frame = step;
              }
            }
            __syncthreads();
            const Se3 pose = s_pose;
            ob_keyframe_points(A, F, s, slot, fo, n, pose, pts, pidx, flag, l, fmax_det, A.occupied + fo, A.po_obs + 3 * fo, step, false, sh);
          }
        }
      }
      __syncthreads();
      int mpn = 0;
      for (int i = tid; i < n; i += OB_T) mpn += (F.mp_valid[fo + i] && F.mp_observed[fo + i]) ? 1 : 0;
      mpn = ob_block_sum(mpn, sh.red);
      if (tid == 0) {
        st->map_points = mpn; st->track_ok = A.track_ok[(size_t)s * K + j]; st->mo_dynamic = O.dyn & 1;
        for (int c = 0; c < 7; c++) { st->tco[c] = O.tco[c]; F.tco[((size_t)s * K + j) * 7 + c] = O.tco[c]; }
      }
    }
    __syncthreads();
    // ---- mLastFrame = Frame(mCurrentFrame): this detection's features ----
    for (int i = tid; i < n; i += OB_T) {
      const size_t o = fo + i;
      L.x[o] = F.x[o]; L.y[o] = F.y[o]; L.angle[o] = F.angle[o]; L.uright[o] = F.uright[o]; L.depth[o] = F.depth[o]; L.octave[o] = F.octave[o];
      L.mp_valid[o] = F.mp_valid[o]; L.mp_observed[o] = F.mp_observed[o]; L.outlier[o] = F.outlier[o]; L.mp_id[o] = F.mp_id[o];
      for (int c = 0; c < 3; c++) L.mp_po[3 * o + c] = F.mp_po[3 * o + c];
    }
    const uint4* sd = reinterpret_cast<const uint4*>(F.desc + fo * 32);
    uint4* dd = reinterpret_cast<uint4*>(L.desc + fo * 32);
    for (int i = tid; i < 2 * n; i += OB_T) dd[i] = sd[i];
    if (tid == 0) {
      L.det[(size_t)s * K + j] = F.det[(size_t)s * K + j];
      L.mo[(size_t)s * K + j] = F.mo[(size_t)s * K + j];
      for (int c = 0; c < 7; c++) L.tco[((size_t)s * K + j) * 7 + c] = F.tco[((size_t)s * K + j) * 7 + c];
    }
  }
  if (j == 0) {
    if (tid <= K) L.off[(size_t)s * (K + 1) + tid] = F.off[(size_t)s * (K + 1) + tid];
    if (tid == 0) L.ndet[s] = nd;
  }
}

// ---- brute-force matcher: the block table of bf_topk derived from the problem table on the device ----
// (compact: only the blocks that hold queries, appended in any order; *count is cleared by the host side before the launch)
__global__ __launch_bounds__(64) void ob_bf_blocks(const BfProb* probs, BfBlock* blocks, int32_t* count, int nprob, int blocks_per_prob) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nprob * blocks_per_prob) return;
  const int p = i / blocks_per_prob, b = i % blocks_per_prob;
  const BfProb P = probs[p];
  const int first = b * PS_BF_QPB;
  if (!(P.nt > 0 && first < P.nq)) return;
  blocks[atomicAdd(count, 1)] = BfBlock{p, first, min(PS_BF_QPB, P.nq - first), 0};
  if (P.nt > PS_BF_SMALL_NT && b == 0) atomicOr(&count[1], 1);   // bf_topk (the large-problem kernel) has work
}

}  // namespace

extern "C" {
void psk_ob_masks(const ObArrays* A, uint8_t* objmask, int W, int H, int ostride, uint8_t* occ, int ocw, int och, hipStream_t st) {
  const size_t WP = (W + 255) & ~255;
  const size_t lds = 3 * WP + 64;
  hipLaunchKernelGGL(ob_masks, dim3(H, A->S), dim3(OB_MASK_T), lds, st, *A, objmask, W, H, ostride, occ, ocw, och);
}
void psk_ob_begin(const ObArrays* A, int step, hipStream_t st) { hipLaunchKernelGGL(ob_begin, dim3(A->S), dim3(OB_T), 0, st, *A, step); }
// (r05) the per-detection kernels are launched sequence-major, (S, K): workgroup b runs on XCD b % 8, and with the detection slot as the fast index -
// (K, S), K = 8 - slot j of EVERY sequence ran on XCD j: two live detections per sequence kept two of the eight XCDs busy and the other six idle
void psk_ob_track(const ObArrays* A, int step, hipStream_t st) { hipLaunchKernelGGL(ob_track, dim3(A->S, A->K), dim3(OB_T), 0, st, *A, step); }
void psk_ob_bf_blocks(const BfProb* probs, BfBlock* blocks, int32_t* count, int nprob, int blocks_per_prob, hipStream_t st) {
  hipMemsetAsync(count, 0, 8, st);
  hipLaunchKernelGGL(ob_bf_blocks, dim3((nprob * blocks_per_prob + 63) / 64), dim3(64), 0, st, probs, blocks, count, nprob, blocks_per_prob);
}
void psk_ob_after_bf(const ObArrays* A, int step, hipStream_t st) { hipLaunchKernelGGL(ob_after_bf, dim3(A->S), dim3(OB_T), 0, st, *A, step); }
void psk_ob_after_cf1(const ObArrays* A, int step, hipStream_t st) { hipLaunchKernelGGL(ob_after_cf1, dim3(A->S, A->K), dim3(OB_T), (size_t)A->LC, st, *A, step); }
void psk_ob_after_lm(const ObArrays* A, int step, hipStream_t st) { hipLaunchKernelGGL(ob_after_lm, dim3(A->S), dim3(OB_T), 0, st, *A, step); }
void psk_ob_finish(const ObArrays* A, int step, hipStream_t st) { hipLaunchKernelGGL(ob_finish, dim3(A->S, A->K), dim3(OB_T), 0, st, *A, step); }
}
