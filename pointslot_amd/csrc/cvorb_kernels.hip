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
}
