// CDNA4 kernels of the object-feature detector: OpenCV's own ORB as Frame::ExtractObjORB calls it
//   cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, kp, descriptor)      /root/reference/src/Frame.cc:2623-2627
// (SURVEY.md 8f-2).  The algorithm lives in un-vendored OpenCV 3.4.x (features2d/src/orb.cpp); it is restated here from its
// published source - INTER_LINEAR_EXACT pyramid, whole-image FAST-9/16 with non-maximum suppression, mask and border filters,
// Harris response, intensity-centroid angle, blurred rBRIEF - and cannot be verified against OpenCV in this image.  The two
// KeyPointsFilter::retainBest steps run on the host between the kernels (cvorb_host.hip): their output ORDER is whatever
// std::nth_element / std::partition leave, so they are executed with those very algorithms.
// Object features are <= ~1000 per image: these kernels are written for clarity first (one thread per pixel / one wave per
// keypoint), they are not on the throughput-critical part of the path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvorb_plan.h"
#include "retain_best.h"

namespace {

__constant__ __attribute__((aligned(16))) int8_t cv_pattern[1024] = {
#include "orb_pattern.inc"
};

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// level 0: copyMakeBorder(image, REFLECT_101) into the padded plane; the mask as it is
__global__ __launch_bounds__(256) void cv_level0(CvLevelDev L, const uint8_t* img, int stride, const uint8_t* mask, int mask_stride) {
  const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
  const int PW = L.w + 2 * CV_BORDER;
  if (px >= PW) return;
  const int x = reflect101(px - CV_BORDER, L.w), y = reflect101(py - CV_BORDER, L.h);
  L.pad[(size_t)py * L.stride + px] = img[(size_t)y * stride + x];
  const int ix = px - CV_BORDER, iy = py - CV_BORDER;
  if (L.mask && ix >= 0 && ix < L.w && iy >= 0 && iy < L.h) L.mask[(size_t)iy * L.w + ix] = mask[(size_t)iy * mask_stride + ix];
}

// resize(prev, cur, INTER_LINEAR_EXACT) (imgproc/src/resize.cpp, resize_bitExact, CV_8UC1): 8.8 coefficients, horizontal pass in
// 8.8, vertical in 16.16, one rounding; tab[d] = {source offset, c0, c1, kind} with kind 0 inside, 1 before the first source
// sample (the first sample alone), 2 after the last.  The padded plane is written in one go: a border pixel is the level pixel at
// the REFLECT_101 coordinate.
__device__ __forceinline__ uint32_t cv_hline(const uint8_t* S, int sw, int4 tx) {
  if (tx.w == 1) return (uint32_t)S[0] << 8;
  if (tx.w == 2) return (uint32_t)S[sw - 1] << 8;
  return (uint32_t)tx.y * S[tx.x] + (uint32_t)tx.z * S[tx.x + 1];
}
__device__ __forceinline__ uint8_t cv_interp(const uint8_t* src, int sstride, int sw, int sh, int4 tx, int4 ty) {
  if (ty.w != 0) {
    const uint32_t v = cv_hline(src + (size_t)(ty.w == 1 ? 0 : sh - 1) * sstride, sw, tx);
    return (uint8_t)min((v + 128u) >> 8, 255u);
  }
  const uint32_t r0 = cv_hline(src + (size_t)ty.x * sstride, sw, tx), r1 = cv_hline(src + (size_t)(ty.x + 1) * sstride, sw, tx);
  const unsigned long long v = (unsigned long long)r0 * (uint32_t)ty.y + (unsigned long long)r1 * (uint32_t)ty.z;
  return (uint8_t)min((v + 32768ull) >> 16, 255ull);
}
__global__ __launch_bounds__(256) void cv_resize(CvLevelDev L, CvLevelDev P, const int4* xtab, const int4* ytab) {
  const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
  const int PW = L.w + 2 * CV_BORDER;
  if (px >= PW) return;
  const int x = reflect101(px - CV_BORDER, L.w), y = reflect101(py - CV_BORDER, L.h);
  const int4 tx = xtab[x], ty = ytab[y];
  const uint8_t* proi = P.pad + (size_t)CV_BORDER * P.stride + CV_BORDER;
  L.pad[(size_t)py * L.stride + px] = cv_interp(proi, P.stride, P.w, P.h, tx, ty);
  const int ix = px - CV_BORDER, iy = py - CV_BORDER;
  if (L.mask && ix >= 0 && ix < L.w && iy >= 0 && iy < L.h) {
    const uint8_t m = cv_interp(P.mask, P.w, P.w, P.h, tx, ty);
    L.mask[(size_t)iy * L.w + ix] = m > 254 ? m : 0;   // threshold(currMask, currMask, 254, 0, THRESH_TOZERO)
  }
}

// FAST-9/16 score plane: s = the largest margin by which 9 contiguous ring pixels are all darker or all brighter than the centre
// (corner at threshold t <=> s > t; cv::FAST reports s - 1), 0 where s <= threshold.  Pixels within 3 of the border are not tested.
__global__ __launch_bounds__(256) void cv_score(CvLevelDev L, int th) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= L.w) return;
  uint8_t out = 0;
  if (x >= 3 && x < L.w - 3 && y >= 3 && y < L.h - 3) {
    const uint8_t* c = L.pad + (size_t)(CV_BORDER + y) * L.stride + CV_BORDER + x;
    const int st = L.stride, v = c[0];
    // two adjacent compass points both darker / brighter: necessary for a 9-arc
    const int n = c[3 * st], e = c[3], so = c[-3 * st], w = c[-3];
    const int M = min(min(max(n, e), max(e, so)), min(max(so, w), max(w, n)));
    const int m = max(max(min(n, e), min(e, so)), max(min(so, w), min(w, n)));
    if (v - M > th || m - v > th) {
      const int off[16] = {3 * st, 3 * st + 1, 2 * st + 2, st + 3, 3, -st + 3, -2 * st + 2, -3 * st + 1,
                           -3 * st, -3 * st - 1, -2 * st - 2, -st - 3, -3, st - 3, 2 * st - 2, 3 * st - 1};
      int r[16];
#pragma unroll
      for (int i = 0; i < 16; i++) r[i] = c[off[i]];
      int best = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        int mx = r[k], mn = r[k];
#pragma unroll
        for (int j = 1; j < 9; j++) { mx = max(mx, r[(k + j) & 15]); mn = min(mn, r[(k + j) & 15]); }
        best = max(best, max(v - mx, mn - v));
      }
      if (best > th) out = (uint8_t)best;
    }
  }
  L.score[(size_t)y * L.w + x] = out;
}

// keypoint predicate of one pixel: strict 3 x 3 maximum of the score plane, inside the mask (KeyPointsFilter::runByPixelsMask)
// and inside the border rectangle (runByImageBorder(edgeThreshold))
__device__ __forceinline__ bool cv_is_keypoint(const CvLevelDev& L, int x, int y, int edge) {
  if (x < edge || x >= L.w - edge || y < edge || y >= L.h - edge) return false;     // (edge >= 3: the score plane is defined there)
  const uint8_t* s = L.score + (size_t)y * L.w + x;
  const int v = s[0];
  if (v == 0) return false;
  const int W = L.w;
  if (!(v > s[-1] && v > s[1] && v > s[-W - 1] && v > s[-W] && v > s[-W + 1] && v > s[W - 1] && v > s[W] && v > s[W + 1])) return false;
  if (L.mask && L.mask[(size_t)y * L.w + x] == 0) return false;     // (int)(pt + 0.5f) of an integer coordinate
  return true;
}

// one wave per image row: counts the row's keypoints
__global__ __launch_bounds__(64) void cv_count(CvLevelDev L, int edge) {
  const int y = blockIdx.x, lane = threadIdx.x;
  int cnt = 0;
  for (int x0 = 0; x0 < L.w; x0 += 64) {
    const int x = x0 + lane;
    cnt += __popcll(__builtin_amdgcn_ballot_w64(x < L.w && cv_is_keypoint(L, x, y, edge)));
  }
  if (lane == 0) L.rowcnt[y] = cnt;
}

// exclusive scan of the row counts (one workgroup; levels have a few hundred rows)
__global__ __launch_bounds__(256) void cv_scan(CvLevelDev L, int32_t* total) {
  __shared__ int part[256];
  const int t = threadIdx.x, per = (L.h + 255) / 256;
  int s = 0;
  for (int i = t * per; i < min((t + 1) * per, L.h); i++) s += L.rowcnt[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) { int acc = 0; for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; } *total = acc; }
  __syncthreads();
  int run = part[t];
  for (int i = t * per; i < min((t + 1) * per, L.h); i++) { const int v = L.rowcnt[i]; L.rowoff[i] = run; run += v; }
}

// HarrisResponses(img, layerinfo, pts, 7, 0.04f) (orb.cpp): Sobel-like gradients over a 7 x 7 block, float combination as written
__device__ __forceinline__ float cv_harris(const CvLevelDev& L, int x0, int y0) {
  const int step = L.stride;
  const uint8_t* ptr0 = L.pad + (size_t)(CV_BORDER + y0 - 3) * step + CV_BORDER + x0 - 3;
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 7; j++) {
      const uint8_t* ptr = ptr0 + i * step + j;
      const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
      const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  const float scale = __fdiv_rn(1.f, __fmul_rn((float)(4 * 7), 255.f));
  const float scale_sq_sq = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  const float sum = __fadd_rn(fa, fb);
  return __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(0.04f, sum), sum)), scale_sq_sq);
}

// raster-ordered emission: (x, y, FAST score - 1, Harris response) per keypoint
__global__ __launch_bounds__(64) void cv_emit(CvLevelDev L, int edge, int cap) {
  const int y = blockIdx.x, lane = threadIdx.x;
  int base = L.rowoff[y];
  for (int x0 = 0; x0 < L.w; x0 += 64) {
    const int x = x0 + lane;
    const bool k = x < L.w && cv_is_keypoint(L, x, y, edge);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(k);
    if (k) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap) L.cand[pos] = make_float4((float)x, (float)y, (float)((int)L.score[(size_t)y * L.w + x] - 1), cv_harris(L, x, y));
    }
    base += __popcll(m);
  }
}

// GaussianBlur(7 x 7, sigma 2, BORDER_REFLECT_101) on CV_8U: 8.8 fixed-point kernel, horizontal pass exact in 16 bits, one
// rounding (x + 2^15) >> 16 (the border pixels come from the padded plane).  OpenCV blurs the level IN PLACE inside its padded
// pyramid buffer: afterwards the level is blurred and the border around it still holds the unblurred REFLECT_101 copies, and
// that is what computeOrbDescriptors reads when a pattern point of a keypoint close to the edge falls outside the level - so the
// output plane is padded too: blurred inside, a copy of the padded level outside.
__global__ __launch_bounds__(256) void cv_blur(CvLevelDev L, int k0, int k1, int k2, int k3) {
  const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
  if (px >= L.w + 2 * CV_BORDER) return;
  const int x = px - CV_BORDER, y = py - CV_BORDER;
  if (x < 0 || x >= L.w || y < 0 || y >= L.h) { L.blur[(size_t)py * L.stride + px] = L.pad[(size_t)py * L.stride + px]; return; }
  const int kq[7] = {k0, k1, k2, k3, k2, k1, k0};
  uint32_t acc = 0;
  for (int j = 0; j < 7; j++) {
    const uint8_t* row = L.pad + (size_t)(py + j - 3) * L.stride + px - 3;
    uint32_t h = 0;
    for (int i = 0; i < 7; i++) h += (uint32_t)kq[i] * row[i];
    acc += (uint32_t)kq[j] * min(h, 65535u);
  }
  L.blur[(size_t)py * L.stride + px] = (uint8_t)min((acc + 32768u) >> 16, 255u);
}

__device__ __forceinline__ float cv_fast_atan2_deg(float y, float x) {   // cv::fastAtan2 (core/src/mathfuncs_core.simd.hpp, atan_f32)
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846), p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846), p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// one wave per selected keypoint: ICAngles, pt *= scale, computeOrbDescriptors (WTA_K = 2) on the blurred level
__global__ __launch_bounds__(64) void cv_describe(CvPlanDev plan, const CvSel* sel, int nsel, ps_keypoint_pod* kps, uint8_t* desc) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= nsel) return;
  const CvSel S = sel[k];
  const CvLevelDev& L = plan.lv[S.level];
  const int x0 = S.x, y0 = S.y;
  const uint8_t* center = L.pad + (size_t)(CV_BORDER + y0) * L.stride + CV_BORDER + x0;
  // intensity centroid over the circular patch: lane v sums row +v and row -v (lane 0: the centre row)
  int m10 = 0, m01 = 0;
  if (lane <= 15) {
    const int v = lane, d = plan.umax[v];
    if (v == 0) {
      for (int u = -15; u <= 15; ++u) m10 += u * center[u];
    } else {
      int v_sum = 0;
      for (int u = -d; u <= d; ++u) {
        const int vp = center[u + v * L.stride], vm = center[u - v * L.stride];
        v_sum += vp - vm;
        m10 += u * (vp + vm);
      }
      m01 = v * v_sum;
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) { m10 += __shfl_xor(m10, dd); m01 += __shfl_xor(m01, dd); }
  const float angle_deg = cv_fast_atan2_deg((float)m01, (float)m10);
  const float px = __fmul_rn((float)x0, L.scale), py = __fmul_rn((float)y0, L.scale);   // kpt.pt *= scale (every level: scale 1 at level 0)
  if (lane == 0) {
    ps_keypoint_pod o;
    o.x = px; o.y = py; o.size = __fmul_rn(31.f, L.scale); o.angle = angle_deg; o.response = S.response; o.octave = S.level; o.class_id = -1;
    kps[k] = o;
  }
  if (lane < 32) {
    const float inv = __fdiv_rn(1.f, L.scale);
    const float angle = __fmul_rn(angle_deg, (float)(3.14159265358979323846 / 180.f));
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* c = L.blur + (size_t)(CV_BORDER + __float2int_rn(__fmul_rn(py, inv))) * L.stride + CV_BORDER + __float2int_rn(__fmul_rn(px, inv));
    const int8_t* pat = cv_pattern + lane * 32;
    int val = 0;
    for (int t = 0; t < 8; t++) {
      int tv[2];
      for (int q = 0; q < 2; q++) {
        const float fx = (float)pat[4 * t + 2 * q], fy = (float)pat[4 * t + 2 * q + 1];
        const float rx = __fsub_rn(__fmul_rn(fx, a), __fmul_rn(fy, b)), ry = __fadd_rn(__fmul_rn(fx, b), __fmul_rn(fy, a));
        tv[q] = c[__float2int_rn(ry) * L.stride + __float2int_rn(rx)];
      }
      val |= (tv[0] < tv[1]) << t;
    }
    desc[(size_t)k * 32 + lane] = (uint8_t)val;
  }
}


// =====================================================================================================================
// Batched, device-resident form: nimg images of one size with object masks (non-zero = object), everything on one stream with
// no host round trip - the ExtractObjORB stage of the device-resident object chain (track_host.hip).  Results are those of
// cv_* above image by image; what differs is the work: only the parts of every level's padded plane that can reach a
// keypoint under the mask are computed (object masks cover a few percent of a frame), and the two retainBest steps run on
// the device with the library's own algorithms (retain_best.h) instead of on the host.
//   cvb_occupancy  8 x 8 cell occupancy of the mask               cvb_plan    needed cells per level -> tile worklists
//   cvb_level0 / cvb_resize   padded planes + mask pyramid        cvb_detect  FAST score + NMS + filters (+ Harris) per tile, in LDS
//   cvb_blur       7 x 7 blur, separable in LDS                   cvb_select  raster order, retainBest x 2
//   cvb_describe   angle + rBRIEF
// Exactness of the restriction (a pixel left out never feeds a kept result).  Planning is done on 8 x 8 cells of the padded planes:
// * a keypoint needs the level mask non-zero at its pixel; the level mask is a chain of 2 x 2 interpolations of the level-0 mask, so
//   it can only be non-zero within 6 (R - 1) + 3 level-0 pixels of an occupied level-0 cell (R = the level's scale): `kp cells`;
// * around a keypoint the pipeline reads at most 22 pixels of the padded plane (rBRIEF reach 19 + blur 3; FAST + NMS 4, Harris 4,
//   IC angle 15): the plane is needed on the kp cells dilated by 3 cells, and, level by level downwards, on every cell of level
//   l - 1 that holds a source pixel of a needed cell of level l; the blur on the kp cells dilated by 3, the FAST score by 1;
// * kernels run on the 32 x 32 tiles that hold a needed cell.  What a tile computes outside the needed cells may come from
//   pixels that were never written; it is never read for a result (keypoints are only taken inside kp cells).
// =====================================================================================================================
__device__ __forceinline__ CvLevelDev cvb_level(const CvbPlan& P, int img, int l) {
  const CvbLevel& B = P.lv[l];
  uint8_t* base = P.arena + (size_t)img * P.arena_pitch;
  CvLevelDev L;
  L.w = B.w; L.h = B.h; L.stride = B.stride; L.scale = B.scale;
  L.pad = base + B.o_pad; L.blur = base + B.o_blur; L.mask = base + B.o_mask; L.score = nullptr;
  L.rowcnt = nullptr; L.rowoff = nullptr; L.cand = nullptr;
  return L;
}

// one workgroup per (image, band of 32 image rows): which 8 x 8 cells hold a non-zero mask pixel
__global__ __launch_bounds__(256) void cvb_occupancy(CvbPlan P, const uint8_t* masks, int mask_stride, size_t mask_pitch) {
  __shared__ uint8_t flag[4][512];
  const int img = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 4 * 512; i += 256) (&flag[0][0])[i] = 0;
  __syncthreads();
  const uint8_t* M = masks + (size_t)img * mask_pitch;
  for (int x = tid; x < P.w0; x += 256)
    for (int b = 0; b < 4; b++) {
      const int y0 = band * 32 + b * 8, y1 = min(y0 + 8, P.h0);
      uint32_t any = 0;
      for (int y = y0; y < y1; y++) any |= M[(size_t)y * mask_stride + x];
      if (any) flag[b][x >> 3] = 1;
    }
  __syncthreads();
  uint8_t* occ = P.occ + (size_t)img * P.ocw * P.och;
  for (int i = tid; i < 4 * P.ocw; i += 256) {
    const int b = i / P.ocw, cx = i % P.ocw, cy = band * 4 + b;
    if (cy < P.och) occ[cy * P.ocw + cx] = flag[b][cx];
  }
}

__device__ __forceinline__ void cvb_reflect_range(int lo, int hi, int len, int& rlo, int& rhi) {
  // image of [lo, hi] under REFLECT_101 on [0, len): union of the three monotone pieces
  rlo = 1 << 30; rhi = -1;
  if (lo < 0) { const int a = -min(hi, 0), b = -lo; rlo = min(rlo, a); rhi = max(rhi, b); }
  if (hi >= 0 && lo <= len - 1) { rlo = min(rlo, max(lo, 0)); rhi = max(rhi, min(hi, len - 1)); }
  if (hi > len - 1) { const int a = 2 * (len - 1) - hi, b = 2 * (len - 1) - max(lo, len - 1); rlo = min(rlo, a); rhi = max(rhi, b); }
  rlo = max(rlo, 0); rhi = min(rhi, len - 1);
}

// appends the tiles of level l that hold a flagged cell to worklist `which`
__device__ __forceinline__ void cvb_append_tiles(const CvbPlan& P, int img, int l, int which, const uint8_t* cells, int grow) {
  const CvbLevel& B = P.lv[l];
  const int nt = B.tw * B.th, tid = threadIdx.x;
  uint32_t* wl = P.wl + ((size_t)which * CV_MAX_LEVELS + l) * P.wl_cap;
  int32_t* cnt = P.wl_count + which * CV_MAX_LEVELS + l;
  for (int t0 = 0; t0 < nt; t0 += 256) {
    const int t = t0 + tid;
    bool a = false;
    if (t < nt) {
      const int tx = t % B.tw, ty = t / B.tw;
      const int cx0 = max(4 * tx - grow, 0), cx1 = min(4 * tx + 3 + grow, B.cw - 1), cy0 = max(4 * ty - grow, 0), cy1 = min(4 * ty + 3 + grow, B.ch - 1);
      for (int cy = cy0; cy <= cy1; cy++)
        for (int cx = cx0; cx <= cx1; cx++) a = a || cells[cy * B.cw + cx];
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(a);
    const int lane = tid & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(cnt, __popcll(m));
    base = __shfl(base, 0);
    if (a) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < P.wl_cap) wl[pos] = ((uint32_t)img << 12) | (uint32_t)t;
    }
  }
}

// one workgroup per image: kp cells and needed cells of every level (dynamic LDS: a byte per cell of every level + two scratch
// planes of the largest level), the three tile worklists per level (0 planes, 1 FAST, 2 blur)
__global__ __launch_bounds__(256) void cvb_plan(CvbPlan P) {
  extern __shared__ uint8_t sm[];
  const int img = blockIdx.x, tid = threadIdx.x;
  uint8_t* kp = sm + P.cell_total;
  uint8_t* tmp = kp + P.cell_max;
  const uint8_t* occ = P.occ + (size_t)img * P.ocw * P.och;
  uint8_t* kpmap = P.kpmap + (size_t)img * P.cell_total;
  for (int l = 0; l < P.nlevels; l++) {
    const CvbLevel& B = P.lv[l];
    const int nc = B.cw * B.ch;
    uint8_t* need = sm + B.cell_off;
    const float Rx = (float)P.w0 / (float)B.w, Ry = (float)P.h0 / (float)B.h;
    const float mx = 6.f * (Rx - 1.f) + 3.f, my = 6.f * (Ry - 1.f) + 3.f;
    for (int c = tid; c < nc; c += 256) {
      const int cx = c % B.cw, cy = c / B.cw;
      int x0 = 8 * cx - CV_BORDER, x1 = x0 + 7, y0 = 8 * cy - CV_BORDER, y1 = y0 + 7;
      x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, B.w - 1); y1 = min(y1, B.h - 1);
      bool any = false;
      if (x0 <= x1 && y0 <= y1) {
        int X0 = (int)floorf((float)x0 * Rx - mx), X1 = (int)ceilf((float)(x1 + 1) * Rx + mx);
        int Y0 = (int)floorf((float)y0 * Ry - my), Y1 = (int)ceilf((float)(y1 + 1) * Ry + my);
        X0 = max(X0, 0) >> 3; Y0 = max(Y0, 0) >> 3; X1 = min(X1, P.w0 - 1) >> 3; Y1 = min(Y1, P.h0 - 1) >> 3;
        for (int oy = Y0; oy <= Y1 && !any; oy++)
          for (int ox = X0; ox <= X1; ox++)
            if (occ[oy * P.ocw + ox]) { any = true; break; }
      }
      kp[c] = any ? 1 : 0;
      kpmap[B.cell_off + c] = any ? 1 : 0;
    }
    __syncthreads();
    cvb_append_tiles(P, img, l, 1, kp, 1);                       // FAST + NMS: kp cells and one cell around them
    for (int c = tid; c < nc; c += 256) {                        // dilation by 3 cells, rows then columns
      const int cx = c % B.cw, cy = c / B.cw;
      uint8_t v = 0;
      for (int d = -3; d <= 3; d++) { const int x = cx + d; if (x >= 0 && x < B.cw) v |= kp[cy * B.cw + x]; }
      tmp[c] = v;
    }
    __syncthreads();
    for (int c = tid; c < nc; c += 256) {
      const int cx = c % B.cw, cy = c / B.cw;
      uint8_t v = 0;
      for (int d = -3; d <= 3; d++) { const int y = cy + d; if (y >= 0 && y < B.ch) v |= tmp[y * B.cw + cx]; }
      need[c] = v;
    }
    __syncthreads();
    cvb_append_tiles(P, img, l, 2, need, 0);                     // blur
    __syncthreads();
  }
  // a level's needed cells need their source pixels one level down
  for (int l = P.nlevels - 1; l >= 1; l--) {
    const CvbLevel& B = P.lv[l];
    const CvbLevel& S = P.lv[l - 1];
    const uint8_t* need = sm + B.cell_off;
    uint8_t* down = sm + S.cell_off;
    const int nc = B.cw * B.ch;
    const double rx = (double)S.w / (double)B.w, ry = (double)S.h / (double)B.h;
    for (int c = tid; c < nc; c += 256) {
      if (!need[c]) continue;
      const int cx = c % B.cw, cy = c / B.cw;
      int lx0, lx1, ly0, ly1;
      cvb_reflect_range(8 * cx - CV_BORDER, min(8 * cx - CV_BORDER + 7, B.w + CV_BORDER - 1), B.w, lx0, lx1);
      cvb_reflect_range(8 * cy - CV_BORDER, min(8 * cy - CV_BORDER + 7, B.h + CV_BORDER - 1), B.h, ly0, ly1);
      int sx0 = (int)floor(((double)lx0 + 0.5) * rx - 0.5) - 1, sx1 = (int)floor(((double)lx1 + 0.5) * rx - 0.5) + 2;
      int sy0 = (int)floor(((double)ly0 + 0.5) * ry - 0.5) - 1, sy1 = (int)floor(((double)ly1 + 0.5) * ry - 0.5) + 2;
      sx0 = max(sx0, 0); sy0 = max(sy0, 0); sx1 = min(sx1, S.w - 1); sy1 = min(sy1, S.h - 1);
      const int ax0 = (sx0 + CV_BORDER) >> 3, ax1 = (sx1 + CV_BORDER) >> 3, ay0 = (sy0 + CV_BORDER) >> 3, ay1 = (sy1 + CV_BORDER) >> 3;
      for (int ay = ay0; ay <= ay1; ay++)
        for (int ax = ax0; ax <= ax1; ax++) down[ay * S.cw + ax] = 1;
    }
    __syncthreads();
  }
  for (int l = 0; l < P.nlevels; l++) cvb_append_tiles(P, img, l, 0, sm + P.lv[l].cell_off, 0);
}

#define CVB_TILE_LOOP(P, which, l)                                                                      \
  const CvbLevel& B = P.lv[l];                                                                          \
  const uint32_t* wl = P.wl + ((size_t)(which) * CV_MAX_LEVELS + (l)) * P.wl_cap;                       \
  const int cnt = min(P.wl_count[(which) * CV_MAX_LEVELS + (l)], P.wl_cap);                             \
  for (int it = blockIdx.x; it < cnt; it += gridDim.x)

__global__ __launch_bounds__(256) void cvb_level0(CvbPlan P, const uint8_t* imgs, int stride, size_t pitch, const uint8_t* masks, int mask_stride, size_t mask_pitch) {
  CVB_TILE_LOOP(P, 0, 0) {
    const uint32_t e = wl[it];
    const int img = (int)(e >> 12), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, 0);
    const uint8_t* I = imgs + (size_t)img * pitch;
    const uint8_t* M = masks + (size_t)img * mask_pitch;
    const int py = CVB_TILE * ty + (threadIdx.x >> 3);
    if (py >= L.h + 2 * CV_BORDER) continue;
    const int y = reflect101(py - CV_BORDER, L.h);
    for (int j = 0; j < 4; j++) {
      const int px = CVB_TILE * tx + (threadIdx.x & 7) * 4 + j;
      if (px >= L.w + 2 * CV_BORDER) break;
      const int x = reflect101(px - CV_BORDER, L.w);
      L.pad[(size_t)py * L.stride + px] = I[(size_t)y * stride + x];
      const int ix = px - CV_BORDER, iy = py - CV_BORDER;
      if (ix >= 0 && ix < L.w && iy >= 0 && iy < L.h) L.mask[(size_t)iy * L.w + ix] = M[(size_t)iy * mask_stride + ix];
    }
  }
}

__global__ __launch_bounds__(256) void cvb_resize(CvbPlan P, int l) {
  CVB_TILE_LOOP(P, 0, l) {
    const uint32_t e = wl[it];
    const int img = (int)(e >> 12), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, l), S = cvb_level(P, img, l - 1);
    const int py = CVB_TILE * ty + (threadIdx.x >> 3);
    if (py >= L.h + 2 * CV_BORDER) continue;
    const int y = reflect101(py - CV_BORDER, L.h);
    const int4 tyv = B.ytab[y];
    const uint8_t* sroi = S.pad + (size_t)CV_BORDER * S.stride + CV_BORDER;
    for (int j = 0; j < 4; j++) {
      const int px = CVB_TILE * tx + (threadIdx.x & 7) * 4 + j;
      if (px >= L.w + 2 * CV_BORDER) break;
      const int x = reflect101(px - CV_BORDER, L.w);
      const int4 txv = B.xtab[x];
      L.pad[(size_t)py * L.stride + px] = cv_interp(sroi, S.stride, S.w, S.h, txv, tyv);
      const int ix = px - CV_BORDER, iy = py - CV_BORDER;
      if (ix >= 0 && ix < L.w && iy >= 0 && iy < L.h) {
        const uint8_t m = cv_interp(S.mask, S.w, S.w, S.h, txv, tyv);
        L.mask[(size_t)iy * L.w + ix] = m > 254 ? m : 0;
      }
    }
  }
}

// FAST score of one pixel from a tile of the padded plane in LDS (row stride TS, c = the pixel): the body of cv_score
template <int TS>
__device__ __forceinline__ uint8_t cvb_fast_score_lds(const uint8_t* c, int th) {
  const int st = TS, v = c[0];
  const int n = c[3 * st], e = c[3], so = c[-3 * st], w = c[-3];
  const int M = min(min(max(n, e), max(e, so)), min(max(so, w), max(w, n)));
  const int m = max(max(min(n, e), min(e, so)), max(min(so, w), min(w, n)));
  if (!(v - M > th || m - v > th)) return 0;
  const int off[16] = {3 * st, 3 * st + 1, 2 * st + 2, st + 3, 3, -st + 3, -2 * st + 2, -3 * st + 1,
                       -3 * st, -3 * st - 1, -2 * st - 2, -st - 3, -3, st - 3, 2 * st - 2, 3 * st - 1};
  int r[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = c[off[i]];
  int best = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    int mx = r[k], mn = r[k];
#pragma unroll
    for (int j = 1; j < 9; j++) { mx = max(mx, r[(k + j) & 15]); mn = min(mn, r[(k + j) & 15]); }
    best = max(best, max(v - mx, mn - v));
  }
  return best > th ? (uint8_t)best : (uint8_t)0;
}

// worklists of all levels in one launch (blockIdx.y = level).  Per tile: the 40 x 40 neighbourhood of the padded plane into LDS,
// FAST scores of the tile and one ring around it (34 x 34) into LDS, then per tile pixel inside a kp cell the keypoint
// predicate of cv_is_keypoint - strict 3 x 3 maximum, mask, border rectangle - and the append with the Harris response
__global__ __launch_bounds__(256) void cvb_detect(CvbPlan P) {
  constexpr int TS = 40, SS = 34;
  __shared__ uint8_t tile[TS * TS];
  __shared__ uint8_t sc[SS * SS];
  const int l = blockIdx.y, tid = threadIdx.x;
  CVB_TILE_LOOP(P, 1, l) {
    const uint32_t e = wl[it];
    const int img = (int)(e >> 12), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, l);
    const int PW = L.w + 2 * CV_BORDER, PH = L.h + 2 * CV_BORDER;
    const int bx = CVB_TILE * tx - 4, by = CVB_TILE * ty - 4;           // padded coordinates of tile[0]
    __syncthreads();
    for (int i = tid; i < TS * TS; i += 256) {
      const int px = bx + i % TS, py = by + i / TS;
      tile[i] = (px >= 0 && px < PW && py >= 0 && py < PH) ? L.pad[(size_t)py * L.stride + px] : 0;
    }
    __syncthreads();
    for (int i = tid; i < SS * SS; i += 256) {
      const int sx = i % SS, sy = i / SS;                                // score pixel: padded (bx + 3 + sx, by + 3 + sy)
      const int x = bx + 3 + sx - CV_BORDER, y = by + 3 + sy - CV_BORDER;
      uint8_t v = 0;
      if (x >= 3 && x < L.w - 3 && y >= 3 && y < L.h - 3) v = cvb_fast_score_lds<TS>(tile + (sy + 3) * TS + sx + 3, P.fast_th);
      sc[i] = v;
    }
    __syncthreads();
    const uint8_t* kpmap = P.kpmap + (size_t)img * P.cell_total + B.cell_off;
    for (int j = 0; j < 4; j++) {
      const int lx = (tid & 7) * 4 + j, ly = tid >> 3;
      const int px = CVB_TILE * tx + lx, py = CVB_TILE * ty + ly;
      const int x = px - CV_BORDER, y = py - CV_BORDER;
      if (x < P.edge || x >= L.w - P.edge || y < P.edge || y >= L.h - P.edge) continue;
      if (!kpmap[(py >> 3) * B.cw + (px >> 3)]) continue;
      const uint8_t* s = sc + (ly + 1) * SS + lx + 1;
      const int v = s[0];
      if (v == 0) continue;
      if (!(v > s[-1] && v > s[1] && v > s[-SS - 1] && v > s[-SS] && v > s[-SS + 1] && v > s[SS - 1] && v > s[SS] && v > s[SS + 1])) continue;
      if (L.mask[(size_t)y * L.w + x] == 0) continue;
      const int slot = img * P.nlevels + l;
      const int pos = atomicAdd(&P.ncand[slot], 1);
      if (pos < CVB_CAND_CAP) P.cand[(size_t)slot * CVB_CAND_CAP + pos] = make_float4((float)x, (float)y, (float)(v - 1), cv_harris(L, x, y));
    }
  }
}

// 7 x 7 blur of a tile: 38 x 38 neighbourhood in LDS, horizontal pass into 16-bit sums, vertical pass, one rounding
__global__ __launch_bounds__(256) void cvb_blur(CvbPlan P) {
  constexpr int TS = 38;
  __shared__ uint8_t tile[TS * TS];
  __shared__ uint16_t hs[TS * 32];
  const int l = blockIdx.y, tid = threadIdx.x;
  const uint32_t kq[7] = {(uint32_t)P.kq[0], (uint32_t)P.kq[1], (uint32_t)P.kq[2], (uint32_t)P.kq[3], (uint32_t)P.kq[2], (uint32_t)P.kq[1], (uint32_t)P.kq[0]};
  CVB_TILE_LOOP(P, 2, l) {
    const uint32_t e = wl[it];
    const int img = (int)(e >> 12), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, l);
    const int PW = L.w + 2 * CV_BORDER, PH = L.h + 2 * CV_BORDER;
    const int bx = CVB_TILE * tx - 3, by = CVB_TILE * ty - 3;
    __syncthreads();
    for (int i = tid; i < TS * TS; i += 256) {
      const int px = bx + i % TS, py = by + i / TS;
      tile[i] = (px >= 0 && px < PW && py >= 0 && py < PH) ? L.pad[(size_t)py * L.stride + px] : 0;
    }
    __syncthreads();
    for (int i = tid; i < TS * 32; i += 256) {
      const int r = i >> 5, c = i & 31;
      const uint8_t* row = tile + r * TS + c;
      uint32_t hsum = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) hsum += kq[k] * row[k];
      hs[i] = (uint16_t)min(hsum, 65535u);
    }
    __syncthreads();
    for (int j = 0; j < 4; j++) {
      const int lx = (tid & 7) * 4 + j, ly = tid >> 3;
      const int px = CVB_TILE * tx + lx, py = CVB_TILE * ty + ly;
      if (px >= PW || py >= PH) continue;
      const int x = px - CV_BORDER, y = py - CV_BORDER;
      uint8_t o;
      if (x < 0 || x >= L.w || y < 0 || y >= L.h) o = tile[(ly + 3) * TS + lx + 3];
      else {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += kq[k] * hs[(ly + k) * 32 + lx];
        o = (uint8_t)min((acc + 32768u) >> 16, 255u);
      }
      L.blur[(size_t)py * L.stride + px] = o;
    }
  }
}

// one workgroup per (image, level): the level's keypoints in raster order (the emission order is arbitrary: bitonic sort by
// (y, x)), then computeKeyPoints' two culls - retainBest(2 * quota) by FAST score, retainBest(quota) by Harris response - run by
// one lane with the library's algorithms (retain_best.h) when a level holds more than its quota
__global__ __launch_bounds__(256) void cvb_select(CvbPlan P) {
  __shared__ uint32_t key[CVB_CAND_CAP];
  __shared__ int32_t idx[CVB_CAND_CAP];
  __shared__ float resp[CVB_CAND_CAP];
  __shared__ int32_t pay[CVB_CAND_CAP];
  __shared__ int nkeep;
  const int slot = blockIdx.x, l = slot % P.nlevels, img = slot / P.nlevels, tid = threadIdx.x;
  const int total = P.ncand[slot];
  const int n = min(total, CVB_CAND_CAP);
  if (total > CVB_CAND_CAP && tid == 0) atomicAdd(&P.overflow[img], 1);
  if (n == 0) { if (tid == 0) P.nsel[slot] = 0; return; }
  const float4* C = P.cand + (size_t)slot * CVB_CAND_CAP;
  int npow = 1;
  while (npow < n) npow <<= 1;
  for (int i = tid; i < npow; i += 256) {
    if (i < n) { const float4 c = C[i]; key[i] = ((uint32_t)c.y << 16) | (uint32_t)c.x; idx[i] = i; }
    else { key[i] = 0xFFFFFFFFu; idx[i] = -1; }
  }
  __syncthreads();
  for (int k = 2; k <= npow; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const bool up = (i & k) == 0;
          const uint32_t a = key[i], b = key[p];
          if ((a > b) == up) { key[i] = b; key[p] = a; const int32_t tt = idx[i]; idx[i] = idx[p]; idx[p] = tt; }
        }
      }
      __syncthreads();
    }
  // list in raster order: response = FAST score, payload = candidate index
  for (int i = tid; i < n; i += 256) { const int c = idx[i]; resp[i] = C[c].z; pay[i] = c; }
  __syncthreads();
  const int quota = P.lv[l].quota;
  RbList Lst{resp, pay};
  if (tid == 0) nkeep = rb_retain_best(Lst, n, 2 * quota);
  __syncthreads();
  const int m1 = nkeep;
  for (int i = tid; i < m1; i += 256) resp[i] = C[pay[i]].w;     // HarrisResponses
  __syncthreads();
  if (tid == 0) nkeep = rb_retain_best(Lst, m1, quota);
  __syncthreads();
  const int m = nkeep;
  CvSel* S = P.sel + (size_t)slot * CVB_CAND_CAP;
  for (int i = tid; i < m; i += 256) { const float4 c = C[pay[i]]; S[i] = CvSel{(int32_t)c.x, (int32_t)c.y, l, resp[i]}; }
  if (tid == 0) P.nsel[slot] = m;
}

// four keypoints per workgroup (one wave each): ICAngles, pt *= scale, computeOrbDescriptors - the body of cv_describe on the
// image's own planes; keypoint k of an image is entry k of the concatenation of its levels' selections
__global__ __launch_bounds__(256) void cvb_describe(CvbPlan P) {
  const int img = blockIdx.y, k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  int l = 0, base = 0, total = 0;
  {
    int acc = 0;
    bool found = false;
    for (int i = 0; i < P.nlevels; i++) {
      const int c = P.nsel[img * P.nlevels + i];
      if (!found && k < acc + c) { l = i; base = acc; found = true; }
      acc += c;
    }
    total = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      P.count[img] = min(total, P.ocap);
      if (total > P.ocap) atomicAdd(&P.overflow[img], 1);
    }
    if (!found || k >= P.ocap) return;
  }
  const CvSel S = P.sel[(size_t)(img * P.nlevels + l) * CVB_CAND_CAP + (k - base)];
  const CvLevelDev L = cvb_level(P, img, l);
  const int x0 = S.x, y0 = S.y;
  const uint8_t* center = L.pad + (size_t)(CV_BORDER + y0) * L.stride + CV_BORDER + x0;
  int m10 = 0, m01 = 0;
  if (lane <= 15) {
    const int v = lane, d = P.umax[v];
    if (v == 0) {
      for (int u = -15; u <= 15; ++u) m10 += u * center[u];
    } else {
      int v_sum = 0;
      for (int u = -d; u <= d; ++u) {
        const int vp = center[u + v * L.stride], vm = center[u - v * L.stride];
        v_sum += vp - vm;
        m10 += u * (vp + vm);
      }
      m01 = v * v_sum;
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) { m10 += __shfl_xor(m10, dd); m01 += __shfl_xor(m01, dd); }
  const float angle_deg = cv_fast_atan2_deg((float)m01, (float)m10);
  const float px = __fmul_rn((float)x0, L.scale), py = __fmul_rn((float)y0, L.scale);
  if (lane == 0) {
    ps_keypoint_pod o;
    o.x = px; o.y = py; o.size = __fmul_rn(31.f, L.scale); o.angle = angle_deg; o.response = S.response; o.octave = S.level; o.class_id = -1;
    P.kps[(size_t)img * P.ocap + k] = o;
  }
  if (lane < 32) {
    const float inv = __fdiv_rn(1.f, L.scale);
    const float angle = __fmul_rn(angle_deg, (float)(3.14159265358979323846 / 180.f));
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* c = L.blur + (size_t)(CV_BORDER + __float2int_rn(__fmul_rn(py, inv))) * L.stride + CV_BORDER + __float2int_rn(__fmul_rn(px, inv));
    const int8_t* pat = cv_pattern + lane * 32;
    int val = 0;
    for (int t = 0; t < 8; t++) {
      int tv[2];
      for (int q = 0; q < 2; q++) {
        const float fx = (float)pat[4 * t + 2 * q], fy = (float)pat[4 * t + 2 * q + 1];
        const float rx = __fsub_rn(__fmul_rn(fx, a), __fmul_rn(fy, b)), ry = __fadd_rn(__fmul_rn(fx, b), __fmul_rn(fy, a));
        tv[q] = c[__float2int_rn(ry) * L.stride + __float2int_rn(rx)];
      }
      val |= (tv[0] < tv[1]) << t;
    }
    P.desc[((size_t)img * P.ocap + k) * 32 + lane] = (uint8_t)val;
  }
}

}  // namespace

extern "C" {
void psk_cv_level0(const CvLevelDev* L, const uint8_t* img, int stride, const uint8_t* mask, int mask_stride, hipStream_t st) {
  hipLaunchKernelGGL(cv_level0, dim3((L->w + 2 * CV_BORDER + 255) / 256, L->h + 2 * CV_BORDER), dim3(256), 0, st, *L, img, stride, mask, mask_stride);
}
void psk_cv_resize(const CvLevelDev* L, const CvLevelDev* P, const int4* xtab, const int4* ytab, hipStream_t st) {
  hipLaunchKernelGGL(cv_resize, dim3((L->w + 2 * CV_BORDER + 255) / 256, L->h + 2 * CV_BORDER), dim3(256), 0, st, *L, *P, xtab, ytab);
}
void psk_cv_detect(const CvLevelDev* L, int th, int edge, int cap, int32_t* total, hipStream_t st) {
  hipLaunchKernelGGL(cv_score, dim3((L->w + 255) / 256, L->h), dim3(256), 0, st, *L, th);
  hipLaunchKernelGGL(cv_count, dim3(L->h), dim3(64), 0, st, *L, edge);
  hipLaunchKernelGGL(cv_scan, dim3(1), dim3(256), 0, st, *L, total);
  hipLaunchKernelGGL(cv_emit, dim3(L->h), dim3(64), 0, st, *L, edge, cap);
}
void psk_cv_blur(const CvLevelDev* L, const int* kq, hipStream_t st) {
  hipLaunchKernelGGL(cv_blur, dim3((L->w + 2 * CV_BORDER + 255) / 256, L->h + 2 * CV_BORDER), dim3(256), 0, st, *L, kq[0], kq[1], kq[2], kq[3]);
}
void psk_cv_describe(const CvPlanDev* plan, const CvSel* sel, int nsel, void* kps, uint8_t* desc, hipStream_t st) {
  if (nsel > 0) hipLaunchKernelGGL(cv_describe, dim3(nsel), dim3(64), 0, st, *plan, sel, nsel, (ps_keypoint_pod*)kps, desc);
}

void psk_cvb_run(const CvbPlan* P, int nimg, const uint8_t* imgs, int stride, size_t pitch, const uint8_t* masks, int mask_stride, size_t mask_pitch,
                 hipStream_t st) {
  const int NL = P->nlevels;
  const int grid = 4096;                                       // persistent loops over the worklists
  hipLaunchKernelGGL(cvb_occupancy, dim3((P->h0 + 31) / 32, nimg), dim3(256), 0, st, *P, masks, mask_stride, mask_pitch);
  const size_t plan_lds = (size_t)P->cell_total + 2 * (size_t)P->cell_max;
  if (plan_lds > 48 * 1024) hipFuncSetAttribute((const void*)cvb_plan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_lds);
  hipLaunchKernelGGL(cvb_plan, dim3(nimg), dim3(256), plan_lds, st, *P);
  hipLaunchKernelGGL(cvb_level0, dim3(grid), dim3(256), 0, st, *P, imgs, stride, pitch, masks, mask_stride, mask_pitch);
  for (int l = 1; l < NL; l++) hipLaunchKernelGGL(cvb_resize, dim3(grid), dim3(256), 0, st, *P, l);
  hipLaunchKernelGGL(cvb_detect, dim3(grid / 4, NL), dim3(256), 0, st, *P);
  hipLaunchKernelGGL(cvb_blur, dim3(grid / 4, NL), dim3(256), 0, st, *P);
  hipLaunchKernelGGL(cvb_select, dim3(nimg * NL), dim3(256), 0, st, *P);
  hipLaunchKernelGGL(cvb_describe, dim3((P->ocap + 3) / 4, nimg), dim3(256), 0, st, *P);
}
}
