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
#include <stdlib.h>
#include "cvorb_plan.h"
#include "retain_best.h"

namespace {

#include "describe_common.h"

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

#ifndef CVB_PLAN_T
#define CVB_PLAN_T 512
#endif
#define CVB_PLAN_MAXCW 1024     // cells per row / per column of a level's padded plane that cvb_plan's row tables hold (host check)
#define CVB_PLAN_MAXCH 512
// one workgroup per image: kp cells and needed cells of every level (dynamic LDS: per level a byte per cell for the plane need and
// one for the mask need, two scratch planes of the largest level, the summed-area table of the occupancy), the three tile
// worklists per level (0 planes, 1 FAST, 2 blur).  An entry of worklist 0 carries bit 31 when the tile also holds cells of the
// mask chain (kp cells and, downwards, the cells that hold their source pixels).  Loops run rows by waves, columns by lanes.
#ifdef PS_CVP_PROFILE   // developer build: 100 MHz ticks per phase of image 0's plan
#define CPP_DECL long long cp_t = wall_clock64(), cp_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CPP_MARK(k) do { const long long _n = wall_clock64(); cp_ph[k] += _n - cp_t; cp_t = _n; } while (0)
#define CPP_PRINT() do { if (threadIdx.x == 0 && blockIdx.x == 0) printf("cvb_plan ticks: sat %lld | cells %lld fast+rowdil %lld coldil %lld blurtiles %lld | down %lld wl0 %lld entries %lld\n", cp_ph[0], cp_ph[1], cp_ph[2], cp_ph[3], cp_ph[4], cp_ph[5], cp_ph[6], cp_ph[7]); } while (0)
#else
#define CPP_DECL
#define CPP_MARK(k)
#define CPP_PRINT()
#endif
__global__ __launch_bounds__(CVB_PLAN_T) void cvb_plan(CvbPlan P) {
  extern __shared__ uint8_t sm[];          // per cell of every level: bit 0 = the plane is needed, bit 1 = the mask is needed (4-byte aligned per level)
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = CVB_PLAN_T / 64;
  CPP_DECL;
  const int cmax4 = (P.cell_max + 3) & ~3;
  uint8_t* kp = sm + ((P.cell_total + 3) & ~3);
  uint8_t* tmp = kp + cmax4;
  uint16_t* sat = reinterpret_cast<uint16_t*>(tmp + cmax4);   // (och + 1) x (ocw + 1), first row / column zero; at an even offset
  uint8_t* tflag = reinterpret_cast<uint8_t*>(sat + (P.ocw + 1) * (P.och + 1));   // per tile of every level: bits 0..2 = the three worklists, bit 3 = mask chain
  __shared__ int s_cnt[3 * CV_MAX_LEVELS], s_base[3 * CV_MAX_LEVELS];
  __shared__ uint32_t s_xr[CVB_PLAN_MAXCW];
  __shared__ uint8_t s_kprow[CVB_PLAN_MAXCH], s_drow[CVB_PLAN_MAXCH];
  for (int i = tid; i < P.tile_total; i += CVB_PLAN_T) tflag[i] = 0;
  if (tid < 3 * CV_MAX_LEVELS) s_cnt[tid] = 0;
  const uint8_t* occ = P.occ + (size_t)img * P.ocw * P.och;
  uint8_t* kpmap = P.kpmap + (size_t)img * P.cell_total;
  const int SW = P.ocw + 1;
  for (int y = wave; y <= P.och; y += NW)
    for (int x = lane; x < SW; x += 64) sat[y * SW + x] = (y > 0 && x > 0) ? occ[(y - 1) * P.ocw + x - 1] : 0;
  __syncthreads();
  for (int y = 1 + tid; y <= P.och; y += CVB_PLAN_T) { uint16_t acc = 0; for (int x = 1; x <= P.ocw; x++) { acc += sat[y * SW + x]; sat[y * SW + x] = acc; } }
  __syncthreads();
  for (int x = 1 + tid; x <= P.ocw; x += CVB_PLAN_T) { uint16_t acc = 0; for (int y = 1; y <= P.och; y++) { acc += sat[y * SW + x]; sat[y * SW + x] = acc; } }
  __syncthreads();
  CPP_MARK(0);
  // Per level: the cells are walked as rows by waves; a row band without an occupied level-0 cell (most rows: the objects cover a part
  // of the image) is cleared with dword stores and skipped by the passes behind it.  The dilations work on four cells per lane.
  for (int l = 0; l < P.nlevels; l++) {
    const CvbLevel& B = P.lv[l];
    uint8_t* need = sm + B.cell_off;
    const int cw4 = B.cw >> 2;
    uint32_t* need4 = reinterpret_cast<uint32_t*>(need);
    uint32_t* kp4 = reinterpret_cast<uint32_t*>(kp);
    uint32_t* tmp4 = reinterpret_cast<uint32_t*>(tmp);
    uint32_t* kpmap4 = reinterpret_cast<uint32_t*>(kpmap + B.cell_off);
    const float Rx = (float)P.w0 / (float)B.w, Ry = (float)P.h0 / (float)B.h;
    const float mx = 6.f * (Rx - 1.f) + 3.f, my = 6.f * (Ry - 1.f) + 3.f;
    // the occupancy columns a cell column can see: X0 | X1 << 16 (0xFFFFFFFF: the column holds no level pixel)
    for (int cx = tid; cx < B.cw; cx += CVB_PLAN_T) {
      const int x0 = max(8 * cx - CV_BORDER, 0), x1 = min(8 * cx - CV_BORDER + 7, B.w - 1);
      uint32_t e = 0xFFFFFFFFu;
      if (x0 <= x1) {
        const int X0 = max((int)floorf((float)x0 * Rx - mx), 0) >> 3, X1 = min((int)ceilf((float)(x1 + 1) * Rx + mx), P.w0 - 1) >> 3;
        e = (uint32_t)X0 | ((uint32_t)X1 << 16);
      }
      s_xr[cx] = e;
    }
    __syncthreads();
    for (int cy = wave; cy < B.ch; cy += NW) {
      const int y0 = max(8 * cy - CV_BORDER, 0), y1 = min(8 * cy - CV_BORDER + 7, B.h - 1);
      bool anyrow = false;
      int Y0 = 0, Y1 = 0, band = 0;
      if (y0 <= y1) {
        Y0 = max((int)floorf((float)y0 * Ry - my), 0) >> 3; Y1 = min((int)ceilf((float)(y1 + 1) * Ry + my), P.h0 - 1) >> 3;
        band = (int)sat[(Y1 + 1) * SW + P.ocw] - (int)sat[Y0 * SW + P.ocw];       // occupied cells anywhere in the band of occupancy rows
      }
      if (band > 0) {
        for (int cx = lane; cx < B.cw; cx += 64) {
          const uint32_t e = s_xr[cx];
          bool any = false;
          if (e != 0xFFFFFFFFu) {
            const int X0 = (int)(e & 0xFFFFu), X1 = (int)(e >> 16);
            const int cnt = (int)sat[(Y1 + 1) * SW + X1 + 1] - (int)sat[Y0 * SW + X1 + 1] - (int)sat[(Y1 + 1) * SW + X0] + (int)sat[Y0 * SW + X0];
            any = cnt > 0;
          }
          const int c = cy * B.cw + cx;
          kp[c] = any ? 1 : 0;
          need[c] = any ? 2 : 0;
          kpmap[B.cell_off + c] = any ? 1 : 0;
          anyrow = anyrow || any;
        }
        anyrow = __any(anyrow);
      } else {
        for (int q = lane; q < cw4; q += 64) { kp4[cy * cw4 + q] = 0; need4[cy * cw4 + q] = 0; kpmap4[cy * cw4 + q] = 0; }
      }
      if (lane == 0) s_kprow[cy] = anyrow ? 1 : 0;
    }
    __syncthreads();
    CPP_MARK(1);
    // FAST + NMS tiles: a tile with a kp cell in its 4 x 4 cells or one cell around them
    for (int ty = wave; ty < B.th; ty += NW) {
      bool rows = false;
      for (int cy = max(4 * ty - 1, 0); cy <= min(4 * ty + 4, B.ch - 1); cy++) rows = rows || s_kprow[cy];
      if (!rows) continue;
      for (int tx = lane; tx < B.tw; tx += 64) {
        uint32_t acc = 0;
        for (int cy = max(4 * ty - 1, 0); cy <= min(4 * ty + 4, B.ch - 1); cy++) {
          const uint32_t cur = kp4[cy * cw4 + tx];
          const uint32_t prv = tx > 0 ? kp4[cy * cw4 + tx - 1] & 0xFF000000u : 0u, nxt = tx + 1 < cw4 ? kp4[cy * cw4 + tx + 1] & 0xFFu : 0u;
          acc |= cur | prv | nxt;
        }
        if (acc) tflag[B.tile_off + ty * B.tw + tx] |= 2u;
      }
    }
    // dilation by 3 cells along the rows: byte j of a dword = OR of the cells j - 3 .. j + 3 (byte-aligned windows of the three dwords around it)
    for (int cy = wave; cy < B.ch; cy += NW) {
      if (!s_kprow[cy]) { for (int q = lane; q < cw4; q += 64) tmp4[cy * cw4 + q] = 0; continue; }
      for (int q = lane; q < cw4; q += 64) {
        const uint32_t cur = kp4[cy * cw4 + q], prv = q > 0 ? kp4[cy * cw4 + q - 1] : 0u, nxt = q + 1 < cw4 ? kp4[cy * cw4 + q + 1] : 0u;
        uint32_t v = cur;
        v |= __builtin_amdgcn_alignbyte(cur, prv, 1u) | __builtin_amdgcn_alignbyte(cur, prv, 2u) | __builtin_amdgcn_alignbyte(cur, prv, 3u);   // cells j - 3, j - 2, j - 1
        v |= __builtin_amdgcn_alignbyte(nxt, cur, 1u) | __builtin_amdgcn_alignbyte(nxt, cur, 2u) | __builtin_amdgcn_alignbyte(nxt, cur, 3u);   // cells j + 1, j + 2, j + 3
        tmp4[cy * cw4 + q] = v;
      }
    }
    __syncthreads();
    CPP_MARK(2);
    // ... and along the columns; kp now holds the dilated set, whose cells also need the plane (bit 0 of need)
    for (int cy = wave; cy < B.ch; cy += NW) {
      bool rows = false;
      for (int y = max(cy - 3, 0); y <= min(cy + 3, B.ch - 1); y++) rows = rows || s_kprow[y];
      if (lane == 0) s_drow[cy] = rows ? 1 : 0;
      if (!rows) { for (int q = lane; q < cw4; q += 64) kp4[cy * cw4 + q] = 0; continue; }
      for (int q = lane; q < cw4; q += 64) {
        uint32_t v = 0;
        for (int y = max(cy - 3, 0); y <= min(cy + 3, B.ch - 1); y++) v |= tmp4[y * cw4 + q];
        kp4[cy * cw4 + q] = v;
        need4[cy * cw4 + q] |= v;
      }
    }
    __syncthreads();
    CPP_MARK(3);
    // blur tiles: the tiles with a cell of the dilated set (level 0: also the tiles of its padded plane, which is only read around
    // its own keypoints - level 1 reads the image)
    for (int ty = wave; ty < B.th; ty += NW) {
      bool rows = false;
      for (int cy = 4 * ty; cy <= min(4 * ty + 3, B.ch - 1); cy++) rows = rows || s_drow[cy];
      if (!rows) continue;
      for (int tx = lane; tx < B.tw; tx += 64) {
        uint32_t acc = 0;
        for (int cy = 4 * ty; cy <= min(4 * ty + 3, B.ch - 1); cy++) acc |= kp4[cy * cw4 + tx];
        if (acc) tflag[B.tile_off + ty * B.tw + tx] |= (l == 0 ? 5u : 4u);
      }
    }
    __syncthreads();
  }
  CPP_MARK(4);
  // a level's needed cells need their source pixels one level down (level 1's come from the image itself)
  for (int l = P.nlevels - 1; l >= 2; l--) {
    const CvbLevel& B = P.lv[l];
    const CvbLevel& S = P.lv[l - 1];
    uint32_t* down = reinterpret_cast<uint32_t*>(sm + S.cell_off);
    const float rx = (float)S.w / (float)B.w, ry = (float)S.h / (float)B.h;
    for (int cy = wave; cy < B.ch; cy += NW) {
      int ly0, ly1;
      cvb_reflect_range(8 * cy - CV_BORDER, min(8 * cy - CV_BORDER + 7, B.h + CV_BORDER - 1), B.h, ly0, ly1);
      const int sy0 = max((int)floorf(((float)ly0 + 0.5f) * ry - 0.5f) - 1, 0), sy1 = min((int)floorf(((float)ly1 + 0.5f) * ry - 0.5f) + 2, S.h - 1);
      const int ay0 = (sy0 + CV_BORDER) >> 3, ay1 = (sy1 + CV_BORDER) >> 3;
      for (int cx = lane; cx < B.cw; cx += 64) {
        const uint32_t bits = sm[B.cell_off + cy * B.cw + cx];
        if (!bits) continue;
        int lx0, lx1;
        cvb_reflect_range(8 * cx - CV_BORDER, min(8 * cx - CV_BORDER + 7, B.w + CV_BORDER - 1), B.w, lx0, lx1);
        const int sx0 = max((int)floorf(((float)lx0 + 0.5f) * rx - 0.5f) - 1, 0), sx1 = min((int)floorf(((float)lx1 + 0.5f) * rx - 0.5f) + 2, S.w - 1);
        const int ax0 = (sx0 + CV_BORDER) >> 3, ax1 = (sx1 + CV_BORDER) >> 3;
        for (int ay = ay0; ay <= ay1; ay++)
          for (int ax = ax0; ax <= ax1; ax++) {
            const int i = ay * S.cw + ax;
            atomicOr(&down[i >> 2], bits << (8 * (i & 3)));
          }
      }
    }
    __syncthreads();
  }
  CPP_MARK(5);
  for (int l = 1; l < P.nlevels; l++) {
    // worklist 0 of the levels above 0: the tiles with a needed cell; bit 3 when one of them is of the mask chain
    const CvbLevel& B = P.lv[l];
    const uint8_t* need = sm + B.cell_off;
    for (int ty = wave; ty < B.th; ty += NW)
      for (int tx = lane; tx < B.tw; tx += 64) {
        uint32_t acc = 0;
        for (int cy = 4 * ty; cy < 4 * ty + 4; cy++) acc |= *reinterpret_cast<const uint32_t*>(need + cy * B.cw + 4 * tx);
        uint8_t f = 0;
        if (acc & 0x03030303u) f |= 1;
        if (acc & 0x02020202u) f |= 8;
        tflag[B.tile_off + ty * B.tw + tx] |= f;
      }
  }
  __syncthreads();
  CPP_MARK(6);
  // worklist entries: positions inside the workgroup by LDS counters, then ONE global atomic per list and level
  constexpr int PER = 8 * 512 / CVB_PLAN_T;                        // tiles per thread (host check: tile_total <= 8 * 512)
  int lpos[PER][3];
  for (int k = 0; k < PER; k++) {
    const int i = tid + k * CVB_PLAN_T;
    lpos[k][0] = lpos[k][1] = lpos[k][2] = -1;
    if (i >= P.tile_total) continue;
    int l = 0;
    for (int q = 1; q < P.nlevels; q++) if (i >= P.lv[q].tile_off) l = q;
    const uint8_t f = tflag[i];
    for (int w = 0; w < 3; w++) if ((f >> w) & 1) lpos[k][w] = atomicAdd(&s_cnt[w * CV_MAX_LEVELS + l], 1);
  }
  __syncthreads();
  if (tid < 3 * CV_MAX_LEVELS) s_base[tid] = s_cnt[tid] > 0 ? atomicAdd(&P.wl_count[tid], s_cnt[tid]) : 0;
  __syncthreads();
  for (int k = 0; k < PER; k++) {
    const int i = tid + k * CVB_PLAN_T;
    if (i >= P.tile_total) continue;
    int l = 0;
    for (int q = 1; q < P.nlevels; q++) if (i >= P.lv[q].tile_off) l = q;
    const uint8_t f = tflag[i];
    for (int w = 0; w < 3; w++)
      if (lpos[k][w] >= 0) {
        const int pos = s_base[w * CV_MAX_LEVELS + l] + lpos[k][w];
        if (pos < P.wl_cap) P.wl[((size_t)w * CV_MAX_LEVELS + l) * P.wl_cap + pos] = ((uint32_t)img << 12) | (uint32_t)(i - P.lv[l].tile_off) | ((w == 0 && (f & 8)) ? 0x80000000u : 0u);
      }
  }
  CPP_MARK(7);
  CPP_PRINT();
}

// the next worklist entry is requested while the current tile is worked on (a tile is a chain of dependent round trips: every one
// taken off the chain counts)
// (the entry is the same for the whole wave: read through the scalar cache - s_load_dword - it lands in an SGPR without the
// vmcnt(0) wait that v_readfirstlane of a vector load puts right behind the load, which made the "prefetch" a round trip per tile.
// The lists were written by the kernel before this one; the scalar cache is invalidated at every kernel start.)
typedef const uint32_t __attribute__((address_space(4))) * cvb_scalar_ptr;
__device__ __forceinline__ uint32_t cvb_sload(const uint32_t* p) { return *(cvb_scalar_ptr)(uintptr_t)p; }
#ifndef CVB_RUN
#define CVB_RUN 1            // consecutive worklist entries (horizontally adjacent tiles, mostly) a wave takes before it jumps by the stride
#endif
#define CVB_TILE_VARS(P, which, l)                                                                      \
  const CvbLevel& B = P.lv[l];                                                                          \
  const uint32_t* wl = P.wl + ((size_t)(which) * CV_MAX_LEVELS + (l)) * P.wl_cap;                       \
  const int cnt = min((int)cvb_sload(reinterpret_cast<const uint32_t*>(P.wl_count) + (which) * CV_MAX_LEVELS + (l)), P.wl_cap); \
  const int xchunk = (cnt + 7) >> 3, xend = min(cnt, ((int)(blockIdx.x & 7) + 1) * xchunk);             \
  const int xstep = (int)(gridDim.x >> 3) * CVB_RUN;                                                    \
  int it = (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) * CVB_RUN, run_i = 0;                \
  auto cvb_next = [&](int i, int r) { return r + 1 < CVB_RUN ? i + 1 : i + xstep - (CVB_RUN - 1); };    \
  uint32_t e_next = it < xend ? cvb_sload(wl + it) : 0u;
#define CVB_TILE_FOR                                                                                    \
  for (uint32_t e = e_next; it < xend && ((e = e_next), (e_next = cvb_next(it, run_i) < xend ? cvb_sload(wl + cvb_next(it, run_i)) : 0u), true); \
       it = cvb_next(it, run_i), run_i = run_i + 1 < CVB_RUN ? run_i + 1 : 0)
#define CVB_HAS_NEXT (cvb_next(it, run_i) < xend)
#define CVB_TILE_LOOP(P, which, l) CVB_TILE_VARS(P, which, l) CVB_TILE_FOR

// XCD-aware order: workgroup b runs on XCD b % 8, each XCD has its own L2, and a worklist keeps the tiles of one image together.
// XCD k therefore takes the k-th eighth of the list front to back: neighbouring tiles - which share the 128-byte lines of their
// halos - meet in one L2 instead of fetching those lines once per XCD (r03 PMC: 4.5 GB of reads per step in cvb_resize before).
// The tile kernels run ONE WAVE per tile (64-thread workgroups): they are chains of dependent memory round trips (worklist entry ->
// tables -> source patch -> result), so what counts is how many tiles are in flight - 32 per CU instead of 8.
#define CVB_TT 64
typedef unsigned short cvb_us2 __attribute__((ext_vector_type(2)));
typedef short cvb_s2 __attribute__((ext_vector_type(2)));
// A workgroup of these kernels is ONE wave: its LDS instructions execute in program order, so lanes exchange data through LDS
// without s_barrier and - what matters - without the vmcnt(0) wait a workgroup barrier brings (the prefetched worklist entry and
// the stores of the previous tile stay in flight).  What is needed is that the compiler keeps the order.
static_assert(CVB_TT == 64, "cvb_wave_sync orders LDS traffic inside ONE wave: the tile kernels must be launched with 64-thread workgroups");
__device__ __forceinline__ void cvb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// one entry of interpolationLinear<ufixedpoint16>::getCoeffs (resize.cpp; the host's exact_table): source index and 8.8 weights of
// destination index val, the same IEEE double operations in the same order; outside [dmin, dmax) the edge sample with weight 256
__device__ __forceinline__ int4 cvb_tab_entry(int val, double scale, int ssize, int dmin, int dmax) {
  if (val < dmin) return make_int4(0, 256, 0, 1);
  if (val >= dmax) return make_int4(ssize - 1, 256, 0, 2);
  const double fval = __dsub_rn(__dmul_rn(scale, __dadd_rn((double)val, 0.5)), 0.5);
  const int ival = (int)floor(fval);
  const int c1 = __double2int_rn(__dmul_rn(__dsub_rn(fval, (double)ival), 256.0));
  return make_int4(ival, 256 - c1, c1, 0);
}

// level 0: copyMakeBorder(image, REFLECT_101) on the tiles around the level's own keypoints (the mask of level 0 is the input)
__global__ __launch_bounds__(CVB_TT) void cvb_level0(CvbPlan P) {
  const int tid = threadIdx.x;
  CVB_TILE_LOOP(P, 0, 0) {
    const int img = (int)((e >> 12) & 0x7FFFFu), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, 0);
    const uint8_t* I = P.imgs + (size_t)img * P.img_pitch;
    const int px0 = CVB_TILE * tx + (tid & 7) * 4;
    int xs[4];
    for (int j = 0; j < 4; j++) xs[j] = reflect101(min(px0 + j, L.w + 2 * CV_BORDER - 1) - CV_BORDER, L.w);
    // (the sixteen byte loads of the lane's four rows first, then the stores: row by row every row was a memory round trip)
    uint32_t v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int py = min(CVB_TILE * ty + (tid >> 3) + 8 * r, L.h + 2 * CV_BORDER - 1);
      const uint8_t* row = I + (size_t)reflect101(py - CV_BORDER, L.h) * P.img_stride;
      v[r] = (uint32_t)row[xs[0]] | ((uint32_t)row[xs[1]] << 8) | ((uint32_t)row[xs[2]] << 16) | ((uint32_t)row[xs[3]] << 24);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int py = CVB_TILE * ty + (tid >> 3) + 8 * r;
      if (py < L.h + 2 * CV_BORDER && px0 < L.stride) *reinterpret_cast<uint32_t*>(L.pad + (size_t)py * L.stride + px0) = v[r];
    }
  }
}

// 16 aligned bytes of GLOBAL memory at an integer address (as a generic pointer the compiler emits flat_load: it counts on the LDS
// counter as well, and every wait for it becomes vmcnt(0) lgkmcnt(0))
typedef uint32_t cvb_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 cvb_gload16(uintptr_t a) {
  const cvb_u4v x = *(const __attribute__((address_space(1))) cvb_u4v*)a;
  return make_uint4(x.x, x.y, x.z, x.w);
}
// level l >= 1 from level l - 1 (level 1 from the image): the tile's table rows and its source patch are staged in LDS, every thread
// interpolates four consecutive pixels of four rows and stores each quad as one dword; the mask likewise on the tiles of the mask chain
__global__ __launch_bounds__(CVB_TT) void cvb_resize(CvbPlan P, int l) {
  constexpr int PS = 48;                          // source patch edge: 32 * 1.2 + taps + rounding
  constexpr int PW = 64;                          // LDS row pitch: four aligned 16-byte chunks cover 48 bytes at any byte offset
  __shared__ __attribute__((aligned(16))) uint8_t patch[PS * PW];
  __shared__ __attribute__((aligned(16))) uint8_t mpatch[PS * PW];
  __shared__ uint8_t shf[PS], mshf[PS];           // byte offset of a row's first source pixel inside its first 16-byte chunk
  __shared__ int4 xt[CVB_TILE], yt[CVB_TILE];     // table entries of the tile's columns / rows, source offsets resolved
  const int tid = threadIdx.x;
  const CvbLevel& Sb = P.lv[l - 1];
  // (the level's constants pinned in scalar registers: left to itself the compiler selects between the two ADDRESSES inside the kernel
  // arguments - the x or the y field - and loads the winner per lane: vector loads, and with them a vmcnt(0) wait in every tile's
  // table step, which also drains the previous tile's stores)
  int dmaxx = P.lv[l].dmaxx, dmaxy = P.lv[l].dmaxy, dminx = P.lv[l].dminx, dminy = P.lv[l].dminy, lvw = P.lv[l].w, lvh = P.lv[l].h, srw = Sb.w, srh = Sb.h;
  double scx = P.lv[l].sx, scy = P.lv[l].sy;
  asm volatile("" : "+s"(dmaxx), "+s"(dmaxy), "+s"(dminx), "+s"(dminy), "+s"(lvw), "+s"(lvh), "+s"(srw), "+s"(srh), "+s"(scx), "+s"(scy));
  CVB_TILE_LOOP(P, 0, l) {
    const bool with_mask = (e >> 31) != 0;
    const int img = (int)((e >> 12) & 0x7FFFFu), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, l);
    const uint8_t* src; int sstride;
    const uint8_t* msrc; int mstride;
    if (l == 1) { src = P.imgs + (size_t)img * P.img_pitch; sstride = P.img_stride; msrc = P.masks + (size_t)img * P.mask_pitch; mstride = P.mask_stride; }
    else {
      const CvLevelDev S = cvb_level(P, img, l - 1);
      src = S.pad + (size_t)CV_BORDER * S.stride + CV_BORDER; sstride = S.stride; msrc = S.mask; mstride = S.w;
    }
    cvb_wave_sync();
    {
      // lanes 0..31: the columns' table entries, lanes 32..63: the rows'; .x = the first source index of the entry
      const int k = tid & 31;
      const bool isx = tid < 32;
      const int len = isx ? lvw : lvh, slen = isx ? srw : srh;
      const int p = min(CVB_TILE * (isx ? tx : ty) + k, len + 2 * CV_BORDER - 1);
      const int4 v = cvb_tab_entry(reflect101(p - CV_BORDER, len), isx ? scx : scy, slen, isx ? dminx : dminy, isx ? dmaxx : dmaxy);
      (isx ? xt : yt)[k] = v;
    }
    cvb_wave_sync();
    // source rectangle: the extreme source indices over the tile's columns / rows (+ 1 for the second tap)
    int ox, oy, nx, ny;
    {
      const int4 v = tid < 32 ? xt[tid & 31] : yt[tid & 31];
      int mn = v.x, mxv = v.w == 0 ? v.x + 1 : v.x;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) { mn = min(mn, __shfl_xor(mn, d)); mxv = max(mxv, __shfl_xor(mxv, d)); }
      ox = __shfl(mn, 0); oy = __shfl(mn, 32);
      nx = min(__shfl(mxv, 0), Sb.w - 1) - ox + 1; ny = min(__shfl(mxv, 32), Sb.h - 1) - oy + 1;
    }
    int mask_mode = 0;                            // 0 no mask on this tile, 1 interpolate, 2 all 255, 3 all 0
    {
      // FOUR lanes per source row (four aligned 16-byte chunks), sixteen rows per pass, three passes: every load is issued before the
      // first LDS store.  A row's first pixel sits `shift` bytes into its first chunk (rows of the image / the tight mask planes
      // start at any byte address).  (r03: with one dword per lane the 24 predicated loads and their address arithmetic were a
      // third of the tile's instructions.)
      uint4 v[PS / 16], mv[PS / 16];
      const int q = tid & 3, r16 = tid >> 2;
      const uintptr_t sb = reinterpret_cast<uintptr_t>(src), mb = reinterpret_cast<uintptr_t>(msrc);
      const uint32_t o0 = __umul24((uint32_t)(oy + r16), (uint32_t)sstride) + (uint32_t)ox, ostep = 16u * (uint32_t)sstride;
      const uint32_t m0 = __umul24((uint32_t)(oy + r16), (uint32_t)mstride) + (uint32_t)ox, mstep = 16u * (uint32_t)mstride;
      uint32_t so = o0, mo = m0;
      uint32_t mloaded = 0;                        // which of this lane's mask chunks were loaded (looked at only after every load is issued)
#pragma unroll
      for (int ry = 0; ry < PS / 16; ry++, so += ostep, mo += mstep) {
        const int yy = r16 + 16 * ry;
        v[ry] = make_uint4(0, 0, 0, 0); mv[ry] = make_uint4(0, 0, 0, 0);
        if (yy < ny) {
          const uintptr_t a = sb + so;
          if (16 * q < (int)(a & 15) + nx) v[ry] = cvb_gload16((a & ~(uintptr_t)15) + 16 * q);
          if (with_mask) {
            const uintptr_t ma = mb + mo;
            if (16 * q < (int)(ma & 15) + nx) { mv[ry] = cvb_gload16((ma & ~(uintptr_t)15) + 16 * q); mloaded |= 1u << ry; }
          }
        }
      }
      // a mask patch that is 255 (inside an object) or 0 (outside) throughout - every chunk that was loaded, the bytes around the
      // patch included - interpolates to that value at every pixel of the tile: only tiles on a mask boundary do the arithmetic
      if (with_mask) {
        bool not255 = false, not0 = false;
#pragma unroll
        for (int ry = 0; ry < PS / 16; ry++)
          if ((mloaded >> ry) & 1u) {
            const uint4 m = mv[ry];
            not255 = not255 || (m.x & m.y & m.z & m.w) != 0xFFFFFFFFu; not0 = not0 || (m.x | m.y | m.z | m.w) != 0u;
          }
        if (!__any(not255)) mask_mode = 2;
        else if (!__any(not0)) mask_mode = 3;
        else mask_mode = 1;
      }
      so = o0; mo = m0;
#pragma unroll
      for (int ry = 0; ry < PS / 16; ry++, so += ostep, mo += mstep) {
        const int yy = r16 + 16 * ry;
        if (yy < ny) {
          *reinterpret_cast<uint4*>(patch + yy * PW + 16 * q) = v[ry];
          if (mask_mode == 1) *reinterpret_cast<uint4*>(mpatch + yy * PW + 16 * q) = mv[ry];
          if (q == 0) {
            shf[yy] = (uint8_t)((sb + so) & 15);
            if (mask_mode == 1) mshf[yy] = (uint8_t)((mb + mo) & 15);
          }
        }
      }
    }
    cvb_wave_sync();
    // cv_interp on the patch.  Both passes carry 8.8 weights that add up to 256 (an edge sample has weight 256, its neighbour 0),
    // so the horizontal sums are at most 255 * 256 (16 bits), the 16.16 total fits 32 bits and never exceeds 255 after the one
    // rounding.  Each pass is one v_dot2_u32_u16 on a packed pair (one instruction instead of two multiplies and an add; all of them issue at the same rate, tools/ubench/int_issue.hip).
    const int c4 = (tid & 7) * 4, px0 = CVB_TILE * tx + c4;
    int cx[4]; cvb_us2 xw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int4 v = xt[c4 + j]; cx[j] = v.x - ox; xw[j] = __builtin_bit_cast(cvb_us2, (uint32_t)v.y | ((uint32_t)v.z << 16)); }
    // r05 (tools/valu_busy.sh: the LDS busy 0.57 of the kernel's time, half of it bank conflicts, the vector ALUs 0.46): the eight source
    // bytes a lane's four pixels can touch - taps min cx .. max cx + 1, at most 8 apart for scale factors up to 2 - come as THREE ALIGNED
    // dwords per source row + v_alignbyte, and v_perm_b32 picks each pixel's (b0, b1) pair as packed u16 (orb_level_fused's form): 4 LDS
    // reads per output row where the byte reads were 16.  (Wider spans - scale factors above 2 - keep the byte reads.)
    // (the columns of a tile are not monotonic where the border reflects: the window starts at the smallest tap)
    const int cmin = min(min(cx[0], cx[1]), min(cx[2], cx[3])), cmax = max(max(cx[0], cx[1]), max(cx[2], cx[3]));
    const bool narrow = __all(cmax - cmin <= 6);
    uint32_t psel[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const uint32_t dj = (uint32_t)(cx[j] - cmin); psel[j] = dj | 0x0c000c00u | ((dj + 1u) << 16); }
    auto interp4 = [&](const uint8_t* rowa, const uint8_t* rowb, cvb_us2 yw, uint32_t* o) {
      if (narrow) {
        const uint8_t* pa = rowa + cmin;
        const uint8_t* pb = rowb + cmin;
        const uint32_t sa = (uint32_t)(uintptr_t)pa & 3u, sb = (uint32_t)(uintptr_t)pb & 3u;
        const uint32_t* qa = reinterpret_cast<const uint32_t*>(pa - sa);
        const uint32_t* qb = reinterpret_cast<const uint32_t*>(pb - sb);
        const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
        const uint32_t al = __builtin_amdgcn_alignbyte(a1, a0, sa), ah = __builtin_amdgcn_alignbyte(a2, a1, sa);
        const uint32_t bl = __builtin_amdgcn_alignbyte(b1, b0, sb), bh = __builtin_amdgcn_alignbyte(b2, b1, sb);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(ah, al, psel[j])), xw[j], 0u, false);
          const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(bh, bl, psel[j])), xw[j], 0u, false);
          o[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(cvb_us2, h0 | (h1 << 16)), yw, 32768u, false) >> 16;
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t wa = (uint32_t)rowa[cx[j]] | ((uint32_t)rowa[cx[j] + 1] << 8), wb = (uint32_t)rowb[cx[j]] | ((uint32_t)rowb[cx[j] + 1] << 8);
        const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(0u, wa, 0x0c010c00u)), xw[j], 0u, false);
        const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(0u, wb, 0x0c010c00u)), xw[j], 0u, false);
        o[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(cvb_us2, h0 | (h1 << 16)), yw, 32768u, false) >> 16;
      }
    };
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ry = (tid >> 3) + 8 * r, py = CVB_TILE * ty + ry;
      if (py >= L.h + 2 * CV_BORDER) continue;
      const int4 tyv = yt[ry];
      const int r0 = tyv.x - oy, r1 = min(r0 + 1, PS - 1);
      const cvb_us2 yw = __builtin_bit_cast(cvb_us2, (uint32_t)tyv.y | ((uint32_t)tyv.z << 16));
      const uint32_t ra = __umul24((uint32_t)r0, (uint32_t)PW), rb = __umul24((uint32_t)r1, (uint32_t)PW);
      uint32_t o[4];
      interp4(patch + ra + shf[r0], patch + rb + shf[r1], yw, o);
      if (px0 < L.stride) *reinterpret_cast<uint32_t*>(L.pad + (__umul24((uint32_t)py, (uint32_t)L.stride) + (uint32_t)px0)) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
      if (mask_mode != 0) {
        if (mask_mode == 1) interp4(mpatch + ra + mshf[r0], mpatch + rb + mshf[r1], yw, o);
        else o[0] = o[1] = o[2] = o[3] = mask_mode == 2 ? 255u : 0u;
        const int iy = py - CV_BORDER, ix0 = px0 - CV_BORDER;
        if (iy >= 0 && iy < L.h) {
          // threshold(currMask, currMask, 254, 0, THRESH_TOZERO).  The mask planes are tight (row stride = level width): the four bytes
          // go out as ONE dword store at whatever alignment the row has (the kernel is bound by memory requests, not by arithmetic)
          uint8_t* mp = L.mask + (__umul24((uint32_t)iy, (uint32_t)L.w) + ix0);
          if (ix0 >= 0 && ix0 + 3 < L.w) {
            const uint32_t pk = (o[0] > 254u ? o[0] : 0u) | ((o[1] > 254u ? o[1] : 0u) << 8) | ((o[2] > 254u ? o[2] : 0u) << 16) | ((o[3] > 254u ? o[3] : 0u) << 24);
            __builtin_memcpy(mp, &pk, 4);
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (ix0 + j >= 0 && ix0 + j < L.w) mp[j] = o[j] > 254u ? (uint8_t)o[j] : (uint8_t)0;
          }
        }
      }
    }
  }
}

// FAST score of one pixel from a tile of the padded plane in LDS (row stride TS, c = the pixel): the largest margin by which 9
// contiguous ring pixels are all darker or all brighter than the centre.  min / max over every window of 9 consecutive ring
// pixels by doubling (windows of 2, 4, 8, then one more): 4 x 16 operations per polarity instead of 8 x 16.
template <int TS>
__device__ __forceinline__ int cvb_fast_score_full(const uint8_t* c) {
  const int st = TS, v = c[0];
  const int off[16] = {3 * st, 3 * st + 1, 2 * st + 2, st + 3, 3, -st + 3, -2 * st + 2, -3 * st + 1,
                       -3 * st, -3 * st - 1, -2 * st - 2, -st - 3, -3, st - 3, 2 * st - 2, 3 * st - 1};
  int r[16], lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = c[off[i]];
#pragma unroll
  for (int i = 0; i < 16; i++) { lo2[i] = min(r[i], r[(i + 1) & 15]); hi2[i] = max(r[i], r[(i + 1) & 15]); }
#pragma unroll
  for (int i = 0; i < 16; i++) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
  int best = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), r[(i + 8) & 15]);
    const int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), r[(i + 8) & 15]);
    best = max(best, max(v - hi9, lo9 - v));
  }
  return best;
}

// HarrisResponses (cv_harris) on a tile of the padded plane in LDS by one wave: c = the keypoint's pixel, row stride TS.  Lanes
// 0..48 take one position of the 7 x 7 block each; the three integer sums (exact, so their order does not matter) by butterfly
template <int TS>
__device__ __forceinline__ float cvb_harris_wave(const uint8_t* c, int lane) {
  int a = 0, b = 0, cc = 0;
  if (lane < 49) {
    const uint8_t* ptr = c + (lane / 7 - 3) * TS + (lane % 7 - 3);
    const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-TS + 1] - ptr[-TS - 1]) + (ptr[TS + 1] - ptr[TS - 1]);
    const int Iy = (ptr[TS] - ptr[-TS]) * 2 + (ptr[TS - 1] - ptr[-TS - 1]) + (ptr[TS + 1] - ptr[-TS + 1]);
    a = Ix * Ix; b = Iy * Iy; cc = Ix * Iy;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); cc += __shfl_xor(cc, d); }
  const float scale = __fdiv_rn(1.f, __fmul_rn((float)(4 * 7), 255.f));
  const float scale_sq_sq = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
  const float fa = (float)a, fb = (float)b, fc = (float)cc;
  const float sum = __fadd_rn(fa, fb);
  return __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(0.04f, sum), sum)), scale_sq_sq);
}

// The same by the 16 lanes of a DPP row, four keypoints per wave at a time: lane l16 takes the positions l16, l16 + 16, l16 + 32 and
// (l16 = 0) 48 of the 7 x 7 block; the integer sums are exact, so the split does not matter.  Every lane of the row returns the response.
template <int TS>
__device__ __forceinline__ float cvb_harris_row(const uint8_t* c, int l16) {
  int a = 0, b = 0, cc = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int pos = l16 + 16 * u;
    if (pos < 49) {
      const uint8_t* ptr = c + (pos / 7 - 3) * TS + (pos % 7 - 3);
      const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-TS + 1] - ptr[-TS - 1]) + (ptr[TS + 1] - ptr[TS - 1]);
      const int Iy = (ptr[TS] - ptr[-TS]) * 2 + (ptr[TS - 1] - ptr[-TS - 1]) + (ptr[TS + 1] - ptr[-TS + 1]);
      a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
    }
  }
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); cc += __shfl_xor(cc, d); }
  const float scale = __fdiv_rn(1.f, __fmul_rn((float)(4 * 7), 255.f));
  const float scale_sq_sq = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
  const float fa = (float)a, fb = (float)b, fc = (float)cc;
  const float sum = __fadd_rn(fa, fb);
  return __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(0.04f, sum), sum)), scale_sq_sq);
}

// worklists of all levels in one launch (blockIdx.y = level).  Per tile: the 40 x 40 neighbourhood of the padded plane into LDS;
// the compass test (two adjacent compass points both darker / brighter: necessary for a 9-arc) over the tile and one ring around
// it (34 x 34) leaves a list of survivors; their exact scores go into the LDS score map; then per tile pixel inside a kp cell the
// keypoint predicate of cv_is_keypoint - strict 3 x 3 maximum, mask, border rectangle - and the append with the Harris response
#ifndef CVB_DET_OCC
#define CVB_DET_OCC 4        // (5 spills: 489 -> 565 us)
#endif
__global__ __launch_bounds__(CVB_TT) __attribute__((amdgpu_waves_per_eu(CVB_DET_OCC, 8))) void cvb_detect(CvbPlan P) {
  constexpr int TS = 40, SS = 34;
  __shared__ __attribute__((aligned(16))) uint8_t tile[TS * TS];
  __shared__ __attribute__((aligned(16))) uint8_t sc[SS * SS + 12];     // 1168 bytes: cleared as 292 dwords
  __shared__ uint16_t surv[SS * SS];
  __shared__ uint16_t kpl[256];                  // the tile's keypoints (NMS leaves at most one per 2 x 2 block)
  __shared__ float hres[CVB_TT];                 // Harris responses of up to 64 of them
  __shared__ uint32_t mtile[CVB_TILE * 8];       // the mask bytes of the tile's 32 x 32 pixels
  __shared__ unsigned long long rowmask[SS];   // per score row: the columns whose score a keypoint cell of this tile can read
  const int l = blockIdx.y, tid = threadIdx.x, th = P.fast_th;
  CVB_TILE_LOOP(P, 1, l) {
    const int img = (int)((e >> 12) & 0x7FFFFu), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, l);
    const int PH = L.h + 2 * CV_BORDER;
    const int bx = CVB_TILE * tx - 4, by = CVB_TILE * ty - 4;           // padded coordinates of tile[0]; bx is a multiple of 4
    const uint8_t* kpmap = P.kpmap + (size_t)img * P.cell_total + B.cell_off;
    const uint8_t* mk = l == 0 ? P.masks + (size_t)img * P.mask_pitch : L.mask;
    const int mks = l == 0 ? P.mask_stride : L.w;
    uint32_t kpm[4], mrow[4];
    cvb_wave_sync();
    // 40 rows of 10 aligned dwords (the plane's row stride is a multiple of 64 and wider than the padded width)
    {
      // (predicated loads: a lane without a dword of the window inside the plane sends no request - with clamped addresses instead of the
      // predicates the kernel was 5 % slower: it is bound by memory requests, not by the branches around them)
      constexpr int NIT = (TS * (TS / 4) + CVB_TT - 1) / CVB_TT;
      uint32_t v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; k++) {            // every load is issued before the first LDS store
        const int i = tid + k * CVB_TT, yy = i / (TS / 4), q = i % (TS / 4);
        const int px = bx + 4 * q, py = by + yy;
        v[k] = (i < TS * (TS / 4) && px >= 0 && px < L.stride && py >= 0 && py < PH) ? *reinterpret_cast<const uint32_t*>(L.pad + (size_t)py * L.stride + px) : 0u;
      }
      // ... and with them what the keypoint predicate reads at this lane's 4 x 4 pixels (four in a row, rows 8 apart): the mask bytes -
      // a second round trip otherwise, for the few pixels that survive the NMS.  A quad that is not wholly inside the level holds no
      // keypoint (they keep `edge` >= 3 pixels from the border): its mask reads as 0 and is not loaded.
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int px = CVB_TILE * tx + (tid & 7) * 4, py = CVB_TILE * ty + (tid >> 3) + 8 * r;
        const int x = px - CV_BORDER, y = py - CV_BORDER;
        uint32_t m = 0;
        if (y >= 0 && y < L.h && x >= 0 && x + 3 < L.w) __builtin_memcpy(&m, mk + (__umul24((uint32_t)y, (uint32_t)mks) + x), 4);
        mrow[r] = m;
      }
      // the tile's 4 x 4 cell flags of the keypoint map: one aligned dword per cell row, the same for every lane (scalar loads)
#pragma unroll
      for (int r = 0; r < 4; r++) kpm[r] = cvb_sload(reinterpret_cast<const uint32_t*>(kpmap + (4 * ty + r) * B.cw + 4 * tx));
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        const int i = tid + k * CVB_TT;
        if (i < TS * (TS / 4)) *reinterpret_cast<uint32_t*>(tile + (i / (TS / 4)) * TS + 4 * (i % (TS / 4))) = v[k];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) mtile[((tid >> 3) + 8 * r) * 8 + (tid & 7)] = mrow[r];
      for (int q = tid; q < (SS * SS + 12) / 4; q += CVB_TT) reinterpret_cast<uint32_t*>(sc)[q] = 0;
    }
    uint32_t kpbits = 0;                          // bit 4 cy + cx: cell (cx, cy) of the tile is a keypoint cell
    {
      // scores are only read on the tile's keypoint cells and one pixel around them: cell (cx, cy) covers score columns 8 cx .. 8 cx + 9
      // and rows 8 cy .. 8 cy + 9 of the 34 x 34 score map.  (Byte cx of kpm[r] is the flag of cell (cx, r).)
      unsigned long long cm[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t f = kpm[r];
        cm[r] = ((f & 0xFFu) ? 0x3FFull : 0) | ((f & 0xFF00u) ? 0x3FFull << 8 : 0) | ((f & 0xFF0000u) ? 0x3FFull << 16 : 0) | ((f & 0xFF000000u) ? 0x3FFull << 24 : 0);
        kpbits |= (uint32_t)(((f & 0xFFu) ? 1u : 0u) | ((f & 0xFF00u) ? 2u : 0u) | ((f & 0xFF0000u) ? 4u : 0u) | ((f & 0xFF000000u) ? 8u : 0u)) << (4 * r);
      }
      // ... restricted to the pixels FAST tests at all: 3 <= x < w - 3, 3 <= y < h - 3 (score pixel (sx, sy) is level pixel (bx + 3 + sx - CV_BORDER, ...))
      const int xb = bx + 3 - CV_BORDER, yb = by + 3 - CV_BORDER;
      const int sx_lo = min(max(3 - xb, 0), SS), sx_hi = min(max(L.w - 3 - xb, 0), SS);
      const unsigned long long colok = ((1ull << sx_hi) - 1ull) & ~((1ull << sx_lo) - 1ull);
      if (tid < SS) {
        const int chi = tid >> 3, clo = (tid - 2) >> 3;                   // the cell rows whose 10-row band holds score row tid
        unsigned long long m = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) if (r == chi || r == clo) m |= cm[r];
        const int y = yb + tid;
        rowmask[tid] = (y >= 3 && y < L.h - 3) ? (m & colok) : 0ull;
      }
    }
    cvb_wave_sync();
    // compass test (two adjacent compass points both darker / brighter: necessary for a 9-arc) over the score pixels that are read,
    // FOUR horizontally adjacent pixels per lane and step on packed u16 (as orb_fast_cells does): "some adjacent pair of compass points
    // is below x" = (N < x or S < x) and (E < x or W < x) = max(min(N, S), min(E, W)) < x.  9 quads per score row (the last one holds
    // two pixels), 306 items in 5 steps; the survivors are numbered by ballot, no LDS counter (their order does not matter: scores go
    // into the score map, keypoints are sorted by cvb_select).  (r04: 18 one-pixel steps before - half of the tile's instructions.)
    int ns = 0;
    {
      constexpr int NQ = (SS + 3) / 4;                                   // 9
      const cvb_s2 thv = {(short)th, (short)th};
#pragma unroll
      for (int k = 0; k < (SS * NQ + CVB_TT - 1) / CVB_TT; k++) {
        const int i = tid + k * CVB_TT, sy = min(i / NQ, SS - 1), q = i - (i / NQ) * NQ;
        const uint32_t rm = i < SS * NQ ? (uint32_t)(rowmask[sy] >> (4 * q)) & 15u : 0u;
        const uint32_t* up = reinterpret_cast<const uint32_t*>(tile + sy * TS + 4 * q);          // row sy: the N / S points of centre row sy + 3
        const uint32_t* ce = reinterpret_cast<const uint32_t*>(tile + (sy + 3) * TS + 4 * q);
        const uint32_t* dn = reinterpret_cast<const uint32_t*>(tile + (sy + 6) * TS + 4 * q);
        const uint32_t n0 = up[0], n1 = up[1], c0 = ce[0], c1 = ce[1], c2 = ce[2], s0 = dn[0], s1 = dn[1];
        bool cand[4];
#pragma unroll
        for (int pq = 0; pq < 2; pq++) {
          // bytes (b, b + 1) of the 8-byte pair {hi, lo} as packed u16; the centre of pixel j sits at byte 3 + j of the row's three dwords
#define PAIR(hi, lo, b) __builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(hi, lo, (uint32_t)(b) | 0x0c000c00u | ((uint32_t)((b) + 1) << 16)))
          const cvb_us2 pN = PAIR(n1, n0, 3 + 2 * pq), pS = PAIR(s1, s0, 3 + 2 * pq), pW = PAIR(c1, c0, 2 * pq);
          const cvb_us2 pE = pq == 0 ? PAIR(c1, c0, 6) : PAIR(c2, c1, 4);
          const cvb_s2 cv = __builtin_bit_cast(cvb_s2, PAIR(c1, c0, 3 + 2 * pq));
#undef PAIR
          const cvb_us2 M = __builtin_elementwise_max(__builtin_elementwise_min(pN, pS), __builtin_elementwise_min(pE, pW));
          const cvb_us2 m = __builtin_elementwise_min(__builtin_elementwise_max(pN, pS), __builtin_elementwise_max(pE, pW));
          const cvb_s2 dd = cv - __builtin_bit_cast(cvb_s2, M);          // > th: two adjacent compass points darker than v - th
          const cvb_s2 bb = __builtin_bit_cast(cvb_s2, m) - cv;          // > th: two adjacent compass points brighter than v + th
          const cvb_s2 mx = __builtin_elementwise_max(dd, bb);
          cand[2 * pq] = mx.x > thv.x && ((rm >> (2 * pq)) & 1u);
          cand[2 * pq + 1] = mx.y > thv.y && ((rm >> (2 * pq + 1)) & 1u);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned long long bm = __ballot(cand[j]);
          if (cand[j]) surv[ns + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = (uint16_t)(sy * SS + 4 * q + j);
          ns += __popcll(bm);
        }
      }
    }
    cvb_wave_sync();
    // exact scores of the survivors, TWO per lane on packed u16 halves: the maximum (dark arcs) and the minimum (bright arcs) of each of
    // the 16 nine-pixel arcs from running maxima / minima of the ring's two 8-blocks - the arc that starts at i < 8 is the block-0
    // suffix from i and the block-1 prefix up to i, the arc that starts at i + 8 wraps the other way (59 packed operations per
    // polarity for two pixels; cvb_fast_score_full, which this restates, takes 176 for one)
    for (int k0 = 0; k0 < ns; k0 += 2 * CVB_TT) {
      int ii[2];
      const uint8_t* c[2];
#pragma unroll
      for (int e2 = 0; e2 < 2; e2++) {
        ii[e2] = surv[min(k0 + CVB_TT * e2 + tid, ns - 1)];            // lanes beyond the end repeat the last entry (their result is dropped)
        const int sy = ii[e2] / SS, sx = ii[e2] - sy * SS;
        c[e2] = tile + (sy + 3) * TS + sx + 3;
      }
      constexpr int st = TS;
      constexpr int off[16] = {3 * st, 3 * st + 1, 2 * st + 2, st + 3, 3, -st + 3, -2 * st + 2, -3 * st + 1,
                               -3 * st, -3 * st - 1, -2 * st - 2, -st - 3, -3, st - 3, 2 * st - 2, 3 * st - 1};
      cvb_us2 x[16];
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = __builtin_bit_cast(cvb_us2, (uint32_t)c[0][off[i]] | ((uint32_t)c[1][off[i]] << 16));
      const cvb_s2 cv = __builtin_bit_cast(cvb_s2, (uint32_t)c[0][0] | ((uint32_t)c[1][0] << 16));
      cvb_us2 S0[8], P0[8], S1[8], P1[8];
      S0[7] = x[7]; P0[0] = x[0]; S1[7] = x[15]; P1[0] = x[8];
#pragma unroll
      for (int i = 6; i >= 0; i--) { S0[i] = __builtin_elementwise_max(x[i], S0[i + 1]); S1[i] = __builtin_elementwise_max(x[8 + i], S1[i + 1]); }
#pragma unroll
      for (int i = 1; i < 8; i++) { P0[i] = __builtin_elementwise_max(x[i], P0[i - 1]); P1[i] = __builtin_elementwise_max(x[8 + i], P1[i - 1]); }
      cvb_us2 lowest_max = __builtin_elementwise_min(__builtin_elementwise_max(S0[0], P1[0]), __builtin_elementwise_max(S1[0], P0[0]));
#pragma unroll
      for (int i = 1; i < 8; i++)
        lowest_max = __builtin_elementwise_min(lowest_max, __builtin_elementwise_min(__builtin_elementwise_max(S0[i], P1[i]), __builtin_elementwise_max(S1[i], P0[i])));
      S0[7] = x[7]; P0[0] = x[0]; S1[7] = x[15]; P1[0] = x[8];
#pragma unroll
      for (int i = 6; i >= 0; i--) { S0[i] = __builtin_elementwise_min(x[i], S0[i + 1]); S1[i] = __builtin_elementwise_min(x[8 + i], S1[i + 1]); }
#pragma unroll
      for (int i = 1; i < 8; i++) { P0[i] = __builtin_elementwise_min(x[i], P0[i - 1]); P1[i] = __builtin_elementwise_min(x[8 + i], P1[i - 1]); }
      cvb_us2 highest_min = __builtin_elementwise_max(__builtin_elementwise_min(S0[0], P1[0]), __builtin_elementwise_min(S1[0], P0[0]));
#pragma unroll
      for (int i = 1; i < 8; i++)
        highest_min = __builtin_elementwise_max(highest_min, __builtin_elementwise_max(__builtin_elementwise_min(S0[i], P1[i]), __builtin_elementwise_min(S1[i], P0[i])));
      const cvb_s2 best = __builtin_elementwise_max(cv - __builtin_bit_cast(cvb_s2, lowest_max), __builtin_bit_cast(cvb_s2, highest_min) - cv);
      if ((int)best.x > th && k0 + tid < ns) sc[ii[0]] = (uint8_t)best.x;
      if ((int)best.y > th && k0 + CVB_TT + tid < ns) sc[ii[1]] = (uint8_t)best.y;
    }
    cvb_wave_sync();
    // keypoints of the tile: the scored survivors inside the tile that pass cv_is_keypoint - strict 3 x 3 maximum, border rectangle,
    // keypoint cell, mask - into an LDS list; then the whole wave on each one's Harris response, then one reservation in the
    // (image, level) candidate list for the tile
    int nk = 0;
    for (int k0 = 0; k0 < ns; k0 += CVB_TT) {
      bool kp = false;
      int q = 0;
      if (k0 + tid < ns) {
        const int i = surv[k0 + tid], sx = i % SS, sy = i / SS, lx = sx - 1, ly = sy - 1;
        if ((unsigned)lx < (unsigned)CVB_TILE && (unsigned)ly < (unsigned)CVB_TILE) {
          const int x = CVB_TILE * tx + lx - CV_BORDER, y = CVB_TILE * ty + ly - CV_BORDER;
          const uint8_t* s = sc + i;
          const int v = s[0];
          kp = v != 0 && x >= P.edge && x < L.w - P.edge && y >= P.edge && y < L.h - P.edge &&
               v > s[-1] && v > s[1] && v > s[-SS - 1] && v > s[-SS] && v > s[-SS + 1] && v > s[SS - 1] && v > s[SS] && v > s[SS + 1] &&
               ((kpbits >> (4 * (ly >> 3) + (lx >> 3))) & 1u) && reinterpret_cast<const uint8_t*>(mtile)[ly * CVB_TILE + lx] != 0;
          q = ly * CVB_TILE + lx;
        }
      }
      const unsigned long long bm = __ballot(kp);
      if (kp) kpl[nk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = (uint16_t)q;
      nk += __popcll(bm);
    }
    cvb_wave_sync();
    if (nk > 0) {
      const int slot = img * P.nlevels + l;
      int base0 = 0;
      if (tid == 0) base0 = atomicAdd(&P.ncand[slot], nk);      // in flight under the Harris responses
      for (int k0 = 0; k0 < nk; k0 += CVB_TT) {
        const int kn = min(CVB_TT, nk - k0);
        // four keypoints per step, one per 16-lane row (r04: the whole wave per keypoint before - 100 instructions each, a third of the tile's)
        for (int k = 0; k < kn; k += 4) {
          const int kk = k + (tid >> 4);
          const int q = kpl[k0 + min(kk, kn - 1)];
          const float hr = cvb_harris_row<TS>(tile + ((q >> 5) + 4) * TS + (q & 31) + 4, tid & 15);
          if ((tid & 15) == 0 && kk < kn) hres[kk] = hr;
        }
        cvb_wave_sync();
        const float mine = tid < kn ? hres[tid] : 0.f;
        cvb_wave_sync();
        const int base = __shfl(base0, 0);
        if (tid < kn && base + k0 + tid < CVB_CAND_CAP) {
          const int q = kpl[k0 + tid], lx = q & 31, ly = q >> 5;
          const int v = sc[(ly + 1) * SS + lx + 1];
          P.cand[(size_t)slot * CVB_CAND_CAP + base + k0 + tid] =
              make_float4((float)(CVB_TILE * tx + lx - CV_BORDER), (float)(CVB_TILE * ty + ly - CV_BORDER), (float)(v - 1), mine);
        }
      }
    }
  }
}

// 7 x 7 blur of a tile: 38 x 38 neighbourhood in LDS (40-byte rows), horizontal pass into 16-bit sums, vertical pass, one rounding.
// GaussianBlur's fixed-point passes are exact integer sums (8.8 weights that add up to 257: a row sum is at most 255 * 257 = 65535,
// the 16.16 total fits 32 bits), so they are taken as dot products: v_dot4_u32_u8 on the row's bytes (the weights are below 256),
// v_dot2_u32_u16 on vertically adjacent row sums.  (r04: 723 -> ~400 VALU instructions per tile; the kernel is bound by them.)
__global__ __launch_bounds__(CVB_TT) void cvb_blur(CvbPlan P) {
  constexpr int TS = 40, TR = 38;
  __shared__ __attribute__((aligned(16))) uint8_t tile[TS * TR];
  __shared__ __attribute__((aligned(16))) uint16_t hs[TR * 32 + 64];
  const int l = blockIdx.y, tid = threadIdx.x;
  const uint32_t k0 = (uint32_t)P.kq[0], k1 = (uint32_t)P.kq[1], k2 = (uint32_t)P.kq[2], k3 = (uint32_t)P.kq[3];
  const uint32_t KA = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), KB = k2 | (k1 << 8) | (k0 << 16);      // taps 0..3, taps 4..6 of a row window
  const cvb_us2 W01 = __builtin_bit_cast(cvb_us2, k0 | (k1 << 16)), W23 = __builtin_bit_cast(cvb_us2, k2 | (k3 << 16)),
                W45 = __builtin_bit_cast(cvb_us2, k2 | (k1 << 16)), W6L = __builtin_bit_cast(cvb_us2, k0), W6H = __builtin_bit_cast(cvb_us2, k0 << 16);
  // The window of the NEXT tile is requested (into registers) before the current one is worked on: a tile is load -> LDS -> two passes ->
  // store, and with the loads of one tile only a wave waited for memory 60 % of its time (r04 PMC: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).
  // No predicates: a dword of the window that lies outside the plane, or beyond the window's last item, is taken from the clamped
  // position instead - what a tile holds outside the plane is never read for a result (the outputs there are the plane's own border
  // pixels; the lanes past the last item repeat it: same value, same LDS address).
  constexpr int NIT = (TR * (TS / 4) + CVB_TT - 1) / CVB_TT;
  uint32_t vn[NIT];
  const int PHl = P.lv[l].h + 2 * CV_BORDER, strl = P.lv[l].stride, twl = P.lv[l].tw;
  auto request = [&](uint32_t ee) {
    const int img = (int)((ee >> 12) & 0x7FFFFu), t = (int)(ee & 4095u), tx = t % twl, ty = t / twl;
    const uint8_t* pad = P.arena + (size_t)img * P.arena_pitch + P.lv[l].o_pad;
    const int bx = CVB_TILE * tx - 4, by = CVB_TILE * ty - 3;
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int i = min(tid + k * CVB_TT, TR * (TS / 4) - 1), yy = i / (TS / 4), q = i % (TS / 4);
      const int px = min(max(bx + 4 * q, 0), strl - 4), py = min(max(by + yy, 0), PHl - 1);
      vn[k] = *reinterpret_cast<const uint32_t*>(pad + (__umul24((uint32_t)py, (uint32_t)strl) + (uint32_t)px));
    }
  };
  CVB_TILE_VARS(P, 2, l)
  if (it < xend) {
    request(e_next);
    // (four stores behind the first request as behind every later one: the compiler's wait for a prefetched dword is the most cautious
    // over the ways into the loop, and coming from here it would otherwise be "everything but the loads behind it" - which, inside the
    // loop, includes the four stores of the tile before)
#pragma unroll
    for (int r = 0; r < 4; r++) *reinterpret_cast<uint32_t*>(P.dump + 256 * r + 4 * tid) = 0u;
  }
  CVB_TILE_FOR {
    const int img = (int)((e >> 12) & 0x7FFFFu), t = (int)(e & 4095u), tx = t % B.tw, ty = t / B.tw;
    const CvLevelDev L = cvb_level(P, img, l);
    const int PH = L.h + 2 * CV_BORDER;
    cvb_wave_sync();
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int i = min(tid + k * CVB_TT, TR * (TS / 4) - 1);
      *reinterpret_cast<uint32_t*>(tile + (i / (TS / 4)) * TS + 4 * (i % (TS / 4))) = vn[k];
    }
    if (CVB_HAS_NEXT) request(e_next);
    cvb_wave_sync();
    // horizontal pass: four adjacent outputs per item from three aligned dwords of the row (bytes 4 q + 1 .. 4 q + 10): output j is
    // the dot product of bytes 1 + j .. 4 + j with taps 0..3 plus that of bytes 5 + j .. 7 + j with taps 4..6
#pragma unroll
    for (int k = 0; k < (TR * 8 + CVB_TT - 1) / CVB_TT; k++) {
      const int i = min(tid + k * CVB_TT, TR * 8 - 1);
      const int r = i >> 3, q = i & 7;
      const uint32_t* rw = reinterpret_cast<const uint32_t*>(tile + r * TS + 4 * q);
      const uint32_t d0 = rw[0], d1 = rw[1], d2 = rw[2];
      uint32_t h[4];
#pragma unroll
      for (int j = 0; j < 3; j++)
        h[j] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, (uint32_t)(1 + j)), KB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, (uint32_t)(1 + j)), KA, 0u, false), false);
      h[3] = __builtin_amdgcn_udot4(d2, KB, __builtin_amdgcn_udot4(d1, KA, 0u, false), false);
      *reinterpret_cast<uint2*>(hs + r * 32 + 4 * q) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    }
    cvb_wave_sync();
    // vertical pass: a 4 x 4 block of outputs per thread from ten rows of four sums.  pr[r][c]: the sums of rows r, r + 1 of column pair c
    // as packed u16 (c = 0: columns 0 / 1 low halves ... see sel), so that two taps are one v_dot2_u32_u16
    {
      const int lx0 = (tid & 7) * 4, ly0 = (tid >> 3) * 4;
      const int px0 = CVB_TILE * tx + lx0;
      uint2 rowv[10];
#pragma unroll
      for (int r = 0; r < 10; r++) rowv[r] = *reinterpret_cast<const uint2*>(hs + (ly0 + r) * 32 + lx0);
      cvb_us2 pr[9][4];
#pragma unroll
      for (int r = 0; r < 9; r++) {
        pr[r][0] = __builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(rowv[r + 1].x, rowv[r].x, 0x05040100u));
        pr[r][1] = __builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(rowv[r + 1].x, rowv[r].x, 0x07060302u));
        pr[r][2] = __builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(rowv[r + 1].y, rowv[r].y, 0x05040100u));
        pr[r][3] = __builtin_bit_cast(cvb_us2, __builtin_amdgcn_perm(rowv[r + 1].y, rowv[r].y, 0x07060302u));
      }
      const int x0 = px0 - CV_BORDER;
      const bool xin = x0 >= 0 && x0 + 3 < L.w;                            // the four columns of this thread lie inside the level
      // The four stores of a thread are unconditional (a row below the plane goes to P.dump; px0 < stride always: the tiles of a row end
      // at or before the row stride): with a store inside a branch the compiler no longer knows how many memory operations follow the
      // prefetched loads and waits for ALL of them - the stores' acknowledgements, a round trip per tile (r04: 397 -> 212 us without stores)
      uint32_t outv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ly = ly0 + r, py = CVB_TILE * ty + ly;
        const int y = py - CV_BORDER;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint32_t acc = __builtin_amdgcn_udot2(pr[r][j], W01, 32768u, false);
          acc = __builtin_amdgcn_udot2(pr[r + 2][j], W23, acc, false);
          acc = __builtin_amdgcn_udot2(pr[r + 4][j], W45, acc, false);
          acc = r < 3 ? __builtin_amdgcn_udot2(pr[r + 6][j], W6L, acc, false) : __builtin_amdgcn_udot2(pr[8][j], W6H, acc, false);
          o[j] = min(acc >> 16, 255u);
        }
        if (!(xin && y >= 0 && y < L.h)) {
          // outside the level (the border of the padded plane): the plane's own pixel
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int x = x0 + j;
            if (x < 0 || x >= L.w || y < 0 || y >= L.h) o[j] = tile[(ly + 3) * TS + lx0 + j + 4];
          }
        }
        outv[r] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int py = CVB_TILE * ty + ly0 + r;
        uint8_t* dst = py < PH ? L.blur + (__umul24((uint32_t)py, (uint32_t)L.stride) + (uint32_t)px0) : P.dump + 256 * r + 4 * tid;
        *reinterpret_cast<uint32_t*>(dst) = outv[r];
      }
    }
  }
}

// one workgroup per (image, level): the level's keypoints in raster order (the emission order is arbitrary: bitonic sort by
// (y, x)), then computeKeyPoints' two culls - retainBest(2 * quota) by FAST score, retainBest(quota) by Harris response - run by
// one lane with the library's algorithms (retain_best.h) when a level holds more than its quota
// Two launches: candidate lists of up to 512 entries (the usual case) run with 8 KB of LDS - twenty workgroups per CU instead of
// five, which is what counts for a kernel whose retainBest steps are one lane's serial walk through LDS -, longer lists with 32 KB
template <int CAP, int LO>
__global__ __launch_bounds__(256) void cvb_select(CvbPlan P, int nslots) {
  __shared__ uint32_t key[CAP];
  __shared__ int32_t idx[CAP];
  __shared__ float resp[CAP];
  __shared__ int32_t pay[CAP];
  __shared__ int nkeep;
  // (the launch for the long lists - rare - is a few workgroups that share all slots: one workgroup per slot, each with 32 KB of LDS
  // to find on a CU before it can start and return, waited 280 us for its turn beside the other lockstep groups' kernels; 5 us alone)
  const int tid = threadIdx.x;
  // LO >= 0 (the launch for the long lists): a workgroup looks at the candidate counts of its share of the slots in parallel and keeps
  // the few it has to work on in `todo`; LO < 0: one workgroup per slot
  __shared__ int todo[1024];
  __shared__ int ntodo;
  int first = blockIdx.x, count = 1;
  if (LO >= 0) {
    const int share = (nslots + (int)gridDim.x - 1) / (int)gridDim.x, s0 = blockIdx.x * share, s1 = min(s0 + share, nslots);
    if (tid == 0) ntodo = 0;
    __syncthreads();
    for (int sl = s0 + tid; sl < s1; sl += 256) {
      const int n = min(P.ncand[sl], CVB_CAND_CAP);
      if (n > LO && n <= CAP) { const int k = atomicAdd(&ntodo, 1); if (k < 1024) todo[k] = sl; }
    }
    __syncthreads();
    count = min(ntodo, 1024);        // (host check: a workgroup's share is at most 1024 slots)
  }
  for (int it = 0; it < count; it++) {
  const int slot = LO >= 0 ? todo[it] : first;
  const int l = slot % P.nlevels, img = slot / P.nlevels;
  const int total = P.ncand[slot];
  const int n = min(total, CVB_CAND_CAP);
  if (n <= LO || n > CAP) continue;                                // the other launch's list (LO = -1: the empty lists too)
  if (total > CVB_CAND_CAP && tid == 0) atomicAdd(&P.overflow[img], 1);
  if (n == 0) { if (tid == 0) P.nsel[slot] = 0; continue; }
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
  // (wave 0 runs the library's selection with its partition loops spread over the lanes, retain_best.h; the sort's key / index arrays serve as its scratch)
  if (tid < 64) { const int k = rbw_retain_best(Lst, n, 2 * quota, reinterpret_cast<int*>(key), idx, tid); if (tid == 0) nkeep = k; }
  __syncthreads();
  const int m1 = nkeep;
  for (int i = tid; i < m1; i += 256) resp[i] = C[pay[i]].w;     // HarrisResponses
  __syncthreads();
  if (tid < 64) { const int k = rbw_retain_best(Lst, m1, quota, reinterpret_cast<int*>(key), idx, tid); if (tid == 0) nkeep = k; }
  __syncthreads();
  const int m = nkeep;
  CvSel* S = P.sel + (size_t)slot * CVB_CAND_CAP;
  for (int i = tid; i < m; i += 256) { const float4 c = C[pay[i]]; S[i] = CvSel{(int32_t)c.x, (int32_t)c.y, l, resp[i]}; }
  if (tid == 0) P.nsel[slot] = m;
  __syncthreads();                                                 // the lists in LDS are free for the next slot
  }
}

// four keypoints per WAVE, one per 16-lane row: ICAngles, pt *= scale, computeOrbDescriptors - the arithmetic of cv_describe on the
// image's own planes; keypoint k of an image is entry k of the concatenation of its levels' selections
#ifndef CVB_DESC_OCC
#define CVB_DESC_OCC 5       // waves per SIMD the register allocation aims at (the LDS admits five workgroups per CU); measured 3 / 4 / 5: 518 / 461 / 435 us
#endif
#ifndef CVB_DESC_T
#define CVB_DESC_T 64        // one wave = four keypoints per workgroup (7.5 KB of LDS): easier to place beside the other lockstep groups' kernels than
#endif                     // sixteen keypoints and 30 KB (256 / 128 / 64 threads: 40.6 / 40.8 / 40.8 k frames/s)
__global__ __launch_bounds__(CVB_DESC_T) __attribute__((amdgpu_waves_per_eu(CVB_DESC_OCC, 8))) void cvb_describe(CvbPlan P, int nimg) {
  // (r05: the body is orb_describe's - describe_common.h - on this detector's planes: the disc's moments from eight unaligned 8-byte loads
  // per lane under byte masks with v_dot4, sin / cos by the [0, 2 pi] routine, the steered pattern on packed floats; before, the moments
  // were 31 byte pairs per lane out of an LDS copy of the disc and sin / cos the general-purpose double routines: 1270 -> see DESIGN.md)
  __shared__ uint4 s_pat[4][16];
  __shared__ uint2 s_icm[8][16];
  __shared__ uint32_t patch_all[CVB_DESC_T / 16][39 * 10];
  for (int i = threadIdx.x; i < 256; i += CVB_DESC_T) {
    reinterpret_cast<uint32_t*>(s_pat)[i] = reinterpret_cast<const uint32_t*>(&c_pattab)[i];
    reinterpret_cast<uint32_t*>(s_icm)[i] = reinterpret_cast<const uint32_t*>(&c_ictab)[i];
  }
  __syncthreads();
  // (r05: one image per XCD at a time - the launch as (8 x blocks per image, images / 8) - cut this kernel's read requests from 11.3 M to 2.8 M per
  // 1024 images and made it SLOWER in the tracker's batches, 1.22 -> 1.45 ms per 3072 images: an object's keypoints crowd a few dozen lines, and 128
  // waves asking one L2 for them at once queue where eight L2s served them side by side.  The image's workgroups stay dealt over all XCDs.)
  const int img = blockIdx.y, bx = (int)blockIdx.x, nbx = (int)gridDim.x;
  (void)nimg;
  const int lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
  int nsel[CV_MAX_LEVELS];
  int total = 0;
  for (int i = 0; i < P.nlevels; i++) { nsel[i] = P.nsel[img * P.nlevels + i]; total += nsel[i]; }
  if (bx == 0 && threadIdx.x == 0) {
    P.count[img] = min(total, P.ocap);
    if (total > P.ocap) atomicAdd(&P.overflow[img], 1);
  }
  total = min(total, P.ocap);
  uint32_t* patch = patch_all[(threadIdx.x >> 6) * 4 + grp];
  const int r4 = l16 >> 2, c4 = l16 & 3;
  for (int k0 = (bx * (CVB_DESC_T / 64) + (threadIdx.x >> 6)) * 4; k0 < total; k0 += nbx * (CVB_DESC_T / 16)) {
    const bool live = k0 + grp < total;
    const int k = live ? k0 + grp : total - 1;
    int l = 0, base = 0;
    {
      int acc = 0;
      for (int i = 0; i < P.nlevels; i++) {
        if (k >= acc && k < acc + nsel[i]) { l = i; base = acc; }
        acc += nsel[i];
      }
    }
    const CvSel S = P.sel[(size_t)(img * P.nlevels + l) * CVB_CAND_CAP + (k - base)];
    const CvLevelDev L = cvb_level(P, img, l);
    const int x0 = S.x, y0 = S.y;
    const float px = __fmul_rn((float)x0, L.scale), py = __fmul_rn((float)y0, L.scale);
    const float inv = __fdiv_rn(1.f, L.scale);
    // ---- the 39 x 39 neighbourhood of the blurred plane that the steered pattern can reach goes to LDS with row-coalesced loads (four
    // lanes x 12 bytes per row, four rows per step, ten steps; issued first, consumed last), shifted to the patch's own first column ----
    const uint8_t* cb = L.blur + (size_t)(CV_BORDER + __float2int_rn(__fmul_rn(py, inv)) - 19) * L.stride + CV_BORDER + __float2int_rn(__fmul_rn(px, inv)) - 19;
    const uint32_t pshift = (uint32_t)(reinterpret_cast<uintptr_t>(cb) & 3);
    const uint32_t sd = (uint32_t)L.stride >> 2;     // row stride in dwords (the planes' strides are multiples of 64)
    uint32_t tmp[10][3];
    {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(cb - pshift) + 3 * c4;
#pragma unroll
      for (int i = 0; i < 10; i++) {
        // row 39 (i = 9, r4 = 3) does not exist in the neighbourhood: that lane repeats row 38 (never read)
        const uint32_t* r = q + (uint32_t)(i < 9 || r4 < 3 ? 4 * i + r4 : 38) * sd;
        tmp[i][0] = r[0]; tmp[i][1] = r[1]; tmp[i][2] = r[2];
      }
    }
    // ---- ICAngles: m10 = sum u I, m01 = sum v I over the disc of the UNBLURRED plane: 32 rows x 4 groups of 8 columns = 128 items, 8 per lane,
    // one unaligned 8-byte load each; disc mask and column weights (u + 15, as bytes) from a table:  m10 = sum (u + 15) I - 15 sum I ----
    int m10, m01 = 0;
    {
      const uint8_t* ca = L.pad + (size_t)(CV_BORDER + y0 - 15 + r4) * L.stride + CV_BORDER + x0 - 15 + 8 * c4;
      uint2 pix[8];
#pragma unroll
      for (int j = 0; j < 8; j++) __builtin_memcpy(&pix[j], ca + (size_t)(4 * j) * L.stride, 8);
      cvb_wave_sync();                               // the previous round's taps are read
#pragma unroll
      for (int i = 0; i < 10; i++) {
        // the dword after the lane's three comes from the next lane of the quad
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tmp[i][0], 0xF9 /* quad_perm [1,2,3,3] */, 0xF, 0xF, false);
        const uint32_t o0 = __builtin_amdgcn_alignbyte(tmp[i][1], tmp[i][0], pshift), o1 = __builtin_amdgcn_alignbyte(tmp[i][2], tmp[i][1], pshift),
                       o2 = __builtin_amdgcn_alignbyte(nxt, tmp[i][2], pshift);
        uint32_t* d = patch + (4 * i + r4) * 10 + 3 * c4;
        if (i < 9 || r4 < 3) {                     // row 39 does not exist
          d[0] = o0;
          if (c4 < 3) { d[1] = o1; d[2] = o2; }    // ten dwords per row: the fourth lane only has the last one
        }
      }
      int s0 = 0;
      uint32_t acc = 0;
      const int v0 = r4 - 15;
      const uint32_t wlo = 0x03020100u + 0x08080808u * (uint32_t)c4, whi = wlo + 0x04040404u;   // u + 15 of the lane's eight columns
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint2 mk = s_icm[j][l16];
        const uint32_t qx = pix[j].x & mk.x, qy = pix[j].y & mk.y;
        const uint32_t sr = __builtin_amdgcn_udot4(qy, 0x01010101u, __builtin_amdgcn_udot4(qx, 0x01010101u, 0u, false), false);
        acc = __builtin_amdgcn_udot4(qy, whi, __builtin_amdgcn_udot4(qx, wlo, acc, false), false);
        s0 += (int)sr;
        m01 += __mul24(v0 + 4 * j, (int)sr);            // (24-bit: a full 32-bit multiply issues at a quarter of the rate)
      }
      m10 = (int)acc - __mul24(15, s0);
    }
    m10 = row_sum_i32(m10);
    m01 = row_sum_i32(m01);
    const float angle_deg = cv_fast_atan2_deg((float)m01, (float)m10);
    if (live && l16 == 0) {
      ps_keypoint_pod o;
      o.x = px; o.y = py; o.size = __fmul_rn(31.f, L.scale); o.angle = angle_deg; o.response = S.response; o.octave = S.level; o.class_id = -1;
      P.kps[(size_t)img * P.ocap + k] = o;
    }
    const float angle = __fmul_rn(angle_deg, (float)(3.14159265358979323846 / 180.f));
    double sn_d, cs_d;
    sincos_0_2pi((double)angle, sn_d, cs_d, c_sincos);
    const float a = (float)cs_d, b = (float)sn_d;
    cvb_wave_sync();
    // A pattern point (x, y) samples the blurred patch at row cvRound(x b + y a), column cvRound(x a - y b), every product and sum rounded to
    // float (orb.cpp computeOrbDescriptors): two floats per instruction, cvRound by adding 1.5 * 2^23 (see orb_describe)
    const ds_f2 ba = {b, a}, anb = {a, -b};
    const unsigned long long ba64 = __builtin_bit_cast(unsigned long long, ba), anb64 = __builtin_bit_cast(unsigned long long, anb);
    const unsigned long long magic64 = 0x4B4000004B400000ull;  // {1.5 * 2^23, 1.5 * 2^23}
    typedef __attribute__((address_space(3))) const uint8_t lds_u8;
    const uint32_t kall = (uint32_t)(uintptr_t)(lds_u8*)reinterpret_cast<const uint8_t*>(patch) + (uint32_t)(19 * 40 + 19) - (0x400000u * 40u + 0x4B400000u);
    uint32_t bits = 0;
#pragma unroll 1
    for (int t4 = 0; t4 < 4; t4++) {
      const uint4 pw4 = s_pat[t4][l16];
#pragma unroll
      for (int tq = 0; tq < 4; tq++) {
        const int tst = 4 * t4 + tq;
        const uint32_t pw = tq == 0 ? pw4.x : tq == 1 ? pw4.y : tq == 2 ? pw4.z : pw4.w;
        // (one packed conversion per point: FP8 bytes -> {x, y} as a float pair; operands as 64-bit integers: register pairs)
        const unsigned long long xy0 = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_cvt_pk_f32_fp8((int)pw, false)),
                                 xy1 = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_cvt_pk_f32_fp8((int)pw, true));
        unsigned long long T0, T1, Q0, Q1;
        asm("v_pk_mul_f32 %0, %4, %6 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %1, %5, %6 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %2, %4, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
            "v_pk_mul_f32 %3, %5, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %2\n\t"
            "v_pk_add_f32 %1, %1, %3\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %8\n\t"
            "v_pk_add_f32 %1, %1, %8\n\t"
            "s_nop 0"
            : "=&v"(T0), "=&v"(T1), "=&v"(Q0), "=&v"(Q1)
            : "v"(xy0), "v"(xy1), "v"(ba64), "v"(anb64), "v"(magic64));
        const uint32_t i0 = (uint32_t)__mul24((int)(uint32_t)T0, 40) + ((uint32_t)(T0 >> 32) + kall);
        const uint32_t i1 = (uint32_t)__mul24((int)(uint32_t)T1, 40) + ((uint32_t)(T1 >> 32) + kall);
        const int t0 = *(lds_u8*)(uintptr_t)i0, t1 = *(lds_u8*)(uintptr_t)i1;
        bits |= (uint32_t)(t0 < t1) << tst;
      }
    }
    if (live) reinterpret_cast<uint16_t*>(P.desc + ((size_t)img * P.ocap + k) * 32)[l16] = (uint16_t)bits;
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

void psk_cvb_run(const CvbPlan* Pin, int nimg, const uint8_t* imgs, int stride, size_t pitch, const uint8_t* masks, int mask_stride, size_t mask_pitch,
                 int occupancy_given, hipStream_t st) {
  CvbPlan Q = *Pin;
  Q.imgs = imgs; Q.img_stride = stride; Q.img_pitch = pitch; Q.masks = masks; Q.mask_stride = mask_stride; Q.mask_pitch = mask_pitch;
  const CvbPlan* P = &Q;
  const int NL = P->nlevels;
  static const int grid_env = getenv("PS_CVB_GRID") ? atoi(getenv("PS_CVB_GRID")) : 0;   // developer knob (a multiple of 32)
  // persistent loops over the worklists, one wave per tile at a time: 32 waves per image, at least 16384, at most 65536 (r05, ms per call of
  // tools/cvorb_batch_bench.py for 4096 / 16384 / 32768 / 65536 waves: 128 images 0.472 / 0.470 / 0.473 / 0.531, 1024 images - a lockstep
  // group of 512 sequences - 2.65 (5632) / 2.48 / 2.31 / 2.29, 2048 images 4.78 / 4.70 / 4.58 / 4.46)
  const int grid = grid_env > 0 ? grid_env : (nimg * 32 < 16384 ? 16384 : nimg * 32 > 65536 ? 65536 : (nimg * 32 + 31) & ~31);
  if (!occupancy_given) hipLaunchKernelGGL(cvb_occupancy, dim3((P->h0 + 31) / 32, nimg), dim3(256), 0, st, *P, masks, mask_stride, mask_pitch);
  const size_t plan_lds = (size_t)((P->cell_total + 3) & ~3) + 2 * (size_t)((P->cell_max + 3) & ~3) + 2 * (size_t)(P->ocw + 1) * (P->och + 1) + (size_t)P->tile_total + 16;
  if (plan_lds > 48 * 1024) hipFuncSetAttribute((const void*)cvb_plan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_lds);
  hipLaunchKernelGGL(cvb_plan, dim3(nimg), dim3(CVB_PLAN_T), plan_lds, st, *P);
  hipLaunchKernelGGL(cvb_level0, dim3(grid), dim3(CVB_TT), 0, st, *P);
  for (int l = 1; l < NL; l++) hipLaunchKernelGGL(cvb_resize, dim3(grid), dim3(CVB_TT), 0, st, *P, l);
  hipLaunchKernelGGL(cvb_detect, dim3(grid / 4, NL), dim3(CVB_TT), 0, st, *P);
  hipLaunchKernelGGL(cvb_blur, dim3(grid / 4, NL), dim3(CVB_TT), 0, st, *P);
  hipLaunchKernelGGL((cvb_select<512, -1>), dim3(nimg * NL), dim3(256), 0, st, *P, nimg * NL);
  {
    const int nslots = nimg * NL, g = nslots < 32 ? nslots : (nslots + 1023) / 1024 > 32 ? (nslots + 1023) / 1024 : 32;   // 32 workgroups, more only so that a share stays within 1024 slots
    hipLaunchKernelGGL((cvb_select<CVB_CAND_CAP, 512>), dim3(g), dim3(256), 0, st, *P, nslots);
  }
  hipLaunchKernelGGL(cvb_describe, dim3(32 * 256 / CVB_DESC_T, nimg), dim3(CVB_DESC_T), 0, st, *P, nimg);
}
}
