// CDNA4 (gfx950) kernels of the ORB front-end.  Wave = 64 lanes everywhere.
//
// Pipeline per batch of images (all images of a batch in one launch):
//   orb_level_fused     x nlevels  one launch per level: bilinear level (OpenCV fixed point), its 19-px REFLECT_101 border and
//                                  its 7x7 sigma-2 blurred plane, the blur taken from LDS
//   orb_fast_cells      x 1        one wave per 30-px FAST cell: compass-point necessary test on packed u16, exact score of the
//                                  survivors (two per lane), per-cell NMS, iniThFAST / minThFAST fallback, bitmap-ranked emission
//   orb_quadtree        x 1 or 2   one workgroup per (level, image): DistributeOctTree as parallel key passes + node-level
//                                  list bookkeeping in LDS (large batches: a second, smaller configuration for the small levels)
//   orb_describe        x 1        four keypoints per wave: intensity-centroid angle + 256-bit rBRIEF from an LDS-staged patch
// (orb_pyramid_level / orb_border / orb_blur are the unfused forms of the first stage, kept behind PS_ORB_FUSED=0.)
//
// Reference semantics each kernel reproduces are cited at the kernel.  Integer stages are exact;
// float expressions use __f*_rn intrinsics so that hipcc cannot contract them into FMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orb_plan.h"

namespace {

// XCD-aware work mapping (MI355X: 8 XCDs, each with a private 4 MB L2; workgroup b is observed to run on XCD b % 8).
// Kernels whose workgroups re-read one image's planes (FAST windows, keypoint patches) keep every image on ONE XCD.
// Workgroups are dealt to the 8 XCDs round-robin in linear order; with a grid of (8 * blocks_per_image, ceil(nimg / 8)) the
// linear index is y * gridDim.x + x and gridDim.x is a multiple of 8, so xcd = x % 8: image = 8 y + x % 8, local block = x / 8
// (no integer division in the kernel).  Placement only affects speed (L2 hit rate), never results.
__device__ __forceinline__ bool xcd_image_block(int nimg, int& img, int& local_block) {
  img = (int)blockIdx.y * 8 + ((int)blockIdx.x & 7);
  local_block = (int)blockIdx.x >> 3;
  return img < nimg;
}
#define PS_XCD_GRID(blocks_per_image, nimg) dim3((unsigned)(blocks_per_image) * 8u, (unsigned)(((nimg) + 7) / 8))

__device__ __forceinline__ int reflect101(int p, int len) {
  // cv::borderInterpolate(BORDER_REFLECT_101); the border (19) is smaller than any level here, but
  // loop anyway so tiny levels stay correct.
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// ------------------------------------------------------------------------------------------------
// Pyramid level: /root/reference/src/ORBextractor.cc:1107-1132 (ComputePyramid) with OpenCV 3.4
// resize(INTER_LINEAR) 8UC1 fixed-point arithmetic and copyMakeBorder(REFLECT_101).
// One thread = 4 consecutive bytes of the padded plane (one dword store).
// xtab[dx] = {sx0, sx1, a0, a1}, ytab[dy] = {sy0, sy1, b0, b1} are built on the host exactly as
// OpenCV builds xofs/ialpha/yofs/ibeta (float maths, saturate_cast<short>(c * 2048)).
// ------------------------------------------------------------------------------------------------
// Interior pixels only; the 19-px REFLECT_101 border of every level is filled afterwards by orb_border
// (resize only reads the interior of the previous level, so the border is not on the cascade's critical path).
// A thread produces 4 consecutive bytes (one dword store) x PYR_ROWS rows.  Source rows are fetched as aligned
// dwords (3 per source row cover the <= 9-byte span of 4 outputs at scale 1.2) and the taps are extracted with
// per-lane byte offsets; row r of level 0 is a shifted copy (2 aligned dwords + v_alignbyte).
#define PYR_ROWS 4
__device__ __forceinline__ uint32_t byte_of3(uint32_t w0, uint32_t w1, uint32_t w2, int off) {
  const uint32_t w = off < 4 ? w0 : (off < 8 ? w1 : w2);
  return (w >> ((off & 3) * 8)) & 0xFFu;
}
// OpenCV 3.4 resize(INTER_LINEAR) coefficient of destination index d for a source of `slen` samples
// (imgproc/src/resize.cpp): f = (float)((d + 0.5) * scale - 0.5) in double, s = floor(f), f -= s, clamped at both
// ends; weights = saturate_cast<short>(cvRound(c * 2048)).  Evaluated on the fly in IEEE double / float (no
// contraction), which is bit-identical to the host-side table OpenCV builds.  clampx: the x axis zeroes the
// fraction at both ends, the y axis only clips the row indices (resize.cpp's `clip`).
struct ResizeTap { int s0, s1, a0, a1; };
__device__ __forceinline__ ResizeTap resize_tap(int d, double scale, int slen, bool is_x) {
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
  int si = (int)floorf(f);
  f = __fsub_rn(f, (float)si);
  ResizeTap t;
  if (is_x) {
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= slen - 1) { f = 0.f; si = slen - 1; }
    t.s0 = si;
    t.s1 = si + 1 < slen ? si + 1 : si;
  } else {
    t.s0 = si >= 0 ? (si < slen ? si : slen - 1) : 0;
    t.s1 = si + 1 >= 0 ? (si + 1 < slen ? si + 1 : slen - 1) : 0;
  }
  t.a0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
  t.a1 = __float2int_rn(__fmul_rn(f, 2048.f));
  return t;
}
template <bool LEVEL0>
__global__ __launch_bounds__(256) void orb_pyramid_level(OrbPlan plan, int level, uint8_t* arena,
                                                        const uint8_t* imgs, int img_stride,
                                                        size_t img_pitch, const int4* tabs) {
  const OrbLevel L = plan.lv[level];
  const int img = blockIdx.z;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int px4 = PS_EDGE - 3 + (blockIdx.x * 64 + threadIdx.x) * 4;   // padded column of the group (multiple of 4)
  const int y0 = (blockIdx.y * 4 + threadIdx.y) * PYR_ROWS;            // level row
  const int x0 = px4 - PS_EDGE;                                        // level column of byte 0 (may be -3..)
  if (x0 >= L.w || y0 >= L.h) return;
  uint8_t* dst = base + L.plane_off + (size_t)PS_EDGE * L.stride + px4;
  int xs[4];                                                           // clamped level columns of the 4 bytes
#pragma unroll
  for (int k = 0; k < 4; k++) xs[k] = min(max(x0 + k, 0), L.w - 1);    // out-of-range bytes are border: rewritten later
  uint32_t packed[PYR_ROWS];
  if (LEVEL0) {
    const uint8_t* src = imgs + (size_t)img * img_pitch;
#pragma unroll
    for (int r = 0; r < PYR_ROWS; r++) {
      const int y = min(y0 + r, L.h - 1);
      const uint8_t* row = src + (size_t)y * img_stride;
      if (x0 >= 0 && x0 + 7 < L.w) {
        const uintptr_t s = reinterpret_cast<uintptr_t>(row + x0);
        const uint32_t* p = reinterpret_cast<const uint32_t*>(s & ~(uintptr_t)3);
        packed[r] = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)(s & 3));
      } else {
        packed[r] = (uint32_t)row[xs[0]] | ((uint32_t)row[xs[1]] << 8) | ((uint32_t)row[xs[2]] << 16) | ((uint32_t)row[xs[3]] << 24);
      }
    }
  } else {
    const OrbLevel S = plan.lv[level - 1];
    const uint8_t* src = base + S.plane_off + (size_t)PS_EDGE * S.stride + PS_EDGE;   // 4-byte aligned + 3
    const double scale_x = 1. / ((double)L.w / S.w), scale_y = 1. / ((double)L.h / S.h);
    int4 tx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const ResizeTap t = resize_tap(xs[k], scale_x, S.w, true);
      tx[k] = make_int4(t.s0, t.s1, t.a0, t.a1);
    }
    const int sx_lo = tx[0].x;
    // aligned window start in the padded source row: source pixel sx sits at byte PS_EDGE + sx of the row
    const int a0 = (PS_EDGE + sx_lo) & ~3;
    int off0[4], off1[4];
    bool fast = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      off0[k] = PS_EDGE + tx[k].x - a0;
      off1[k] = PS_EDGE + tx[k].y - a0;
      fast = fast && off0[k] >= 0 && off1[k] < 12;
    }
    const uint8_t* srow0 = base + S.plane_off + (size_t)PS_EDGE * S.stride;          // padded row start, 64-B aligned
#pragma unroll
    for (int r = 0; r < PYR_ROWS; r++) {
      const ResizeTap tyy = resize_tap(min(y0 + r, L.h - 1), scale_y, S.h, false);
      const int4 ty = make_int4(tyy.s0, tyy.s1, tyy.a0, tyy.a1);
      uint32_t pk = 0;
      if (fast) {
        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(srow0 + (size_t)ty.x * S.stride + a0);
        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(srow0 + (size_t)ty.y * S.stride + a0);
        const uint32_t u0 = q0[0], u1 = q0[1], u2 = q0[2], v0 = q1[0], v1 = q1[1], v2 = q1[2];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int h0 = (int)byte_of3(u0, u1, u2, off0[k]) * tx[k].z + (int)byte_of3(u0, u1, u2, off1[k]) * tx[k].w;
          const int h1 = (int)byte_of3(v0, v1, v2, off0[k]) * tx[k].z + (int)byte_of3(v0, v1, v2, off1[k]) * tx[k].w;
          int o = (((ty.z * (h0 >> 4)) >> 16) + ((ty.w * (h1 >> 4)) >> 16) + 2) >> 2;
          o = o < 0 ? 0 : (o > 255 ? 255 : o);
          pk |= (uint32_t)o << (8 * k);
        }
      } else {
        const uint8_t* r0 = src + (size_t)ty.x * S.stride;
        const uint8_t* r1 = src + (size_t)ty.y * S.stride;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int h0 = (int)r0[tx[k].x] * tx[k].z + (int)r0[tx[k].y] * tx[k].w;
          const int h1 = (int)r1[tx[k].x] * tx[k].z + (int)r1[tx[k].y] * tx[k].w;
          int o = (((ty.z * (h0 >> 4)) >> 16) + ((ty.w * (h1 >> 4)) >> 16) + 2) >> 2;
          o = o < 0 ? 0 : (o > 255 ? 255 : o);
          pk |= (uint32_t)o << (8 * k);
        }
      }
      packed[r] = pk;
    }
  }
#pragma unroll
  for (int r = 0; r < PYR_ROWS; r++)
    if (y0 + r < L.h) *reinterpret_cast<uint32_t*>(dst + (size_t)(y0 + r) * L.stride) = packed[r];
}

// ------------------------------------------------------------------------------------------------
// Fused level kernel: one launch per level produces the PADDED plane (interior + 19-px REFLECT_101 border) and the
// level's GaussianBlur plane (ORBextractor.cc:1107-1132 + :1085-1086), replacing orb_pyramid_level + orb_border + orb_blur.
//   * The output domain is the padded plane; a padded pixel (px, py) is the level pixel (reflect101(px - 19),
//     reflect101(py - 19)), so the border is just more resize evaluations - no second pass, no read-back.
//   * A workgroup owns 248 x 50 padded pixels and evaluates a 256 x 56 region (4 / 3 pixels of halo) into LDS;
//     the 7x7 blur of the owned in-image pixels is then taken from LDS, so the plane is never re-read from HBM.
//   * Arithmetic: the 8 source bytes that the 4 pixels of a lane can touch are brought into two registers
//     (3 aligned dword loads + v_alignbyte); v_perm_b32 extracts each pixel's (b0, b1) pair as packed u16 and
//     v_dot2_u32_u16 forms b0*a0 + b1*a1 in one instruction; the vertical step is two v_mul_hi_u32 (the
//     weights are pre-shifted by 16) and one add3.  No saturation is needed: a0 + a1 = 2048 bounds the result by 255.
// ------------------------------------------------------------------------------------------------
// bytes {sat8(a >> 16), sat8(b >> 16), sat8(c >> 16), sat8(d >> 16)} of four sums below 2^31 (sat8: values above 255 become 255)
__device__ __forceinline__ uint32_t ps_sat_pack4_hi16(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x07060302u), cd = __builtin_amdgcn_perm(d, c, 0x07060302u);   // {a.hi16, b.hi16}, {c.hi16, d.hi16}
  uint32_t lo, hi;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(lo) : "v"(ab));
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(hi) : "v"(cd));
  return lo | (hi << 16);
}
#ifndef LV_RPT
#define LV_RPT 14                 // region rows per thread (r05: 14 rows in two groups of 7 - 56-row regions, 50 owned: less halo and 13 + 6 instead of
#endif                            // 9 + 6 rows in a thread's blur chunk; ORB 6.36 -> 6.27 ms per 512 sequences; 12: 6.35, 8: 6.49, 16 spills)
#ifndef LV_WAVES
#define LV_WAVES 4                // waves per workgroup: each takes LV_RPT region rows
#endif
#define LV_R (LV_WAVES * LV_RPT)  // region rows per workgroup
#define LV_OWN_R (LV_R - 6)       // owned rows
#define LV_OWN_C 248              // owned columns (62 dword groups; lanes 0 and 63 are halo)
#ifndef LV_HALVES
#define LV_HALVES 2            // the region rows of a thread are loaded and resized in this many groups (with LV_RPT 10: 1 -> 43.85, 2 -> 43.6, 5 -> 43.4 k frames/s)
#endif
#define LV_HROWS (LV_RPT / LV_HALVES)
#define LV_BLUR_ROWS ((LV_OWN_R + LV_WAVES - 1) / LV_WAVES)   // LV_WAVES row chunks cover the owned rows (13 of 50 with 4 waves)
typedef unsigned short lv_us2 __attribute__((ext_vector_type(2)));

#ifdef PS_LV_PROFILE   // developer build: 100 MHz ticks of one workgroup in the middle of the launch
#define LVP_DECL long long lv_t[8]; int lv_n = 0; lv_t[lv_n++] = wall_clock64()
#define LVP_MARK() (lv_t[lv_n++] = wall_clock64())
#define LVP_PRINT() do { if (threadIdx.x == 0 && bx == 2 && by == 5 && img == 64) printf("level %d ticks: setup %lld half0 %lld half1 %lld barrier %lld blur %lld\n", level, lv_t[1] - lv_t[0], lv_t[2] - lv_t[1], lv_t[3] - lv_t[2], lv_t[4] - lv_t[3], wall_clock64() - lv_t[4]); } while (0)
#else
#define LVP_DECL
#define LVP_MARK()
#define LVP_PRINT()
#endif
#ifndef PS_LV_WAVES
#define PS_LV_WAVES 7        // 70 VGPRs: seven waves per SIMD (measured best of 4 / 6 / 7 / 8)
#endif
template <bool LEVEL0>
__global__ __launch_bounds__(64 * LV_WAVES) __attribute__((amdgpu_waves_per_eu(PS_LV_WAVES, 8))) void orb_level_fused(OrbPlan plan, int level, uint8_t* arena, const uint8_t* imgs,
                                                      int img_stride, size_t img_pitch, const int4* tabs, int nimg, int tiles_x) {
  __shared__ uint32_t tile[LV_R + 2][64];   // (+ 2: the last blur chunk reads two rows past the region for outputs it never stores - no clamp, the rows are immediate offsets)
  LVP_DECL;
  const OrbLevel L = plan.lv[level];
  // one image per XCD at a time, its tiles in row-major order: tiles that share halo rows and 128-byte lines meet in one L2
  int img, lb;
  if (!xcd_image_block(nimg, img, lb)) return;
  const int bx = lb % tiles_x, by = lb / tiles_x;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int lane = threadIdx.x & 63;
  const int tyq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int PW = L.w + 2 * PS_EDGE, PH = L.h + 2 * PS_EDGE;
  const int P0 = bx * LV_OWN_C - 4;      // padded column of region column 0 (multiple of 4)
  const int Q0 = by * LV_OWN_R - 3;      // padded row of region row 0

  // ---- phase 1: resize (or copy) the region into LDS, store the owned part of the padded plane ----
  int s0[4], s1[4];
  uint32_t coef[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = min(max(P0 + 4 * lane + k, 0), PW - 1);
    const int lx = reflect101(px - PS_EDGE, L.w);
    if (LEVEL0) { s0[k] = lx; s1[k] = lx; coef[k] = 2048u; }
    else {
      const int4 t = tabs[L.xtab_off + lx];
      s0[k] = t.x; s1[k] = t.y; coef[k] = (uint32_t)t.z | ((uint32_t)t.w << 16);
    }
  }
  // first byte of the lane's 8-byte source window: the smallest tap; at level 0 - whose rows are the caller's, with nothing behind the
  // last one - pulled back so that the window ends inside the row (r05: lanes at the image's left / right edge took the byte gather
  // before, and with them, since the window test is per wave, every wave of the two edge tiles of a tile row)
  const int sb = LEVEL0 ? min(min(min(s0[0], s0[1]), min(s0[2], s0[3])), L.w - 8) : min(min(s0[0], s0[1]), min(s0[2], s0[3]));
  uint32_t sel[4];
  bool fast = !LEVEL0 || L.w >= 8;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i0 = s0[k] - sb, i1 = s1[k] - sb;
    fast = fast && i0 < 8 && i1 < 8;
    sel[k] = (uint32_t)i0 | (0x0cu << 8) | ((uint32_t)i1 << 16) | (0x0cu << 24);
  }
  const uint32_t sel_copy = (uint32_t)(s0[0] - sb) | ((uint32_t)(s0[1] - sb) << 8) | ((uint32_t)(s0[2] - sb) << 16) | ((uint32_t)(s0[3] - sb) << 24);
  // source pixel (x, y) lives at sbase + srel + y * src_stride + x: a wave-uniform 64-bit base plus 32-bit offsets, so the
  // loads are "scalar base + lane offset" and the per-row address arithmetic stays on the scalar unit
  const uint8_t* sbase;
  uint32_t srel, src_stride;
  if (LEVEL0) {
    sbase = imgs + (size_t)img * img_pitch; srel = 0; src_stride = (uint32_t)img_stride;
  } else {
    const OrbLevel S = plan.lv[level - 1];
    sbase = base; srel = S.plane_off + (uint32_t)(PS_EDGE * S.stride + PS_EDGE); src_stride = (uint32_t)S.stride;
  }
  const bool fast_wave = __all(fast);
  const bool own_x = lane >= 1 && lane <= 62 && P0 + 4 * lane < PW;
  uint8_t* plane = base + L.plane_off;
  // Consecutive output rows share a source row four times out of five at scale 1.2 (s0 of row r is s1 of row r - 1), so
  // the horizontally interpolated row is carried over instead of being recomputed; the row indices are wave-uniform.
  uint32_t hc[4] = {0, 0, 0, 0};
  int hc_row = -1;
#ifdef PS_LV_PROFILE
  if (coef[0] == 0x7fffffffu) tile[0][0] = 1;   // keeps the table loads before the first stamp
  LVP_MARK();
#endif
  // the row table entries of all the wave's rows are requested here, beside the column entries above: one round trip instead of
  // one per half in front of the rows' own loads (r03 phase timers: a workgroup spends 11 - 16 of its 17 microseconds in the halves)
  int4 ty_all[LV_RPT];
#pragma unroll
  for (int r = 0; r < LV_RPT; r++) {
    const int py = min(max(Q0 + tyq * LV_RPT + r, 0), PH - 1);
    const int ly = reflect101(py - PS_EDGE, L.h);
    if (LEVEL0) ty_all[r] = make_int4(ly, ly, 0, 0);
    else ty_all[r] = tabs[L.ytab_off + ly];
  }
#pragma unroll
  for (int half = 0; half < LV_HALVES; half++) {
    uint32_t wl[LV_HROWS][2], wh[LV_HROWS][2];
    int4 ty[LV_HROWS];
    bool need0[LV_HROWS];
#pragma unroll
    for (int r = 0; r < LV_HROWS; r++) {
      ty[r] = ty_all[half * (LV_HROWS) + r];
      const int prev_s1 = r > 0 ? ty[r - 1].y : hc_row;
      need0[r] = LEVEL0 || ty[r].x != prev_s1;
    }
    if (fast_wave) {
      // every lane's eight source bytes are one window (all of levels >= 1, level 0 away from the image's left / right edge): ONE
      // 8-byte load per lane and row at the byte address itself (global loads take any alignment), all the half's rows requested
      // before the first is looked at.  (r05: three aligned dwords + v_alignbyte inside the lane-dependent `fast` test made the
      // compiler close every row's block with s_waitcnt vmcnt(0) - six or seven memory round trips per half, one after the other.)
#pragma unroll
      for (int r = 0; r < LV_HROWS; r++)
#pragma unroll
        for (int v = 0; v < (LEVEL0 ? 1 : 2); v++) {
          if (v == 0 && !need0[r]) continue;
          const uint32_t roff = srel + (uint32_t)(v == 0 ? ty[r].x : ty[r].y) * src_stride;   // wave-uniform
          uint2 w;
          __builtin_memcpy(&w, sbase + (roff + (uint32_t)sb), 8);
          wl[r][v] = w.x; wh[r][v] = w.y;
        }
    } else {
#pragma unroll
    for (int r = 0; r < LV_HROWS; r++) {
#pragma unroll
      for (int v = 0; v < (LEVEL0 ? 1 : 2); v++) {
        if (v == 0 && !need0[r]) continue;
        const uint32_t roff = srel + (uint32_t)(v == 0 ? ty[r].x : ty[r].y) * src_stride;   // wave-uniform
        if (fast) {
          uint2 w;
          __builtin_memcpy(&w, sbase + (roff + (uint32_t)sb), 8);
          wl[r][v] = w.x; wh[r][v] = w.y;
        } else {
          // generic gather (image edges of level 0, scale factors above 2): byte k <- s0[k], byte 4 + k <- s1[k]
          // (32-bit offsets from the wave-uniform base: as 64-bit lane addresses the eight sign-extended indices held 16 registers for the whole phase)
#define LV_SB(i) ((uint32_t)sbase[roff + (uint32_t)(i)])
          wl[r][v] = LV_SB(s0[0]) | (LV_SB(s0[1]) << 8) | (LV_SB(s0[2]) << 16) | (LV_SB(s0[3]) << 24);
          wh[r][v] = LV_SB(s1[0]) | (LV_SB(s1[1]) << 8) | (LV_SB(s1[2]) << 16) | (LV_SB(s1[3]) << 24);
#undef LV_SB
        }
      }
    }
    }
#pragma unroll
    for (int r = 0; r < LV_HROWS; r++) {
      const int rr = tyq * LV_RPT + half * (LV_HROWS) + r;
      uint32_t pk;
      if (LEVEL0) {
        pk = fast ? __builtin_amdgcn_perm(wh[r][0], wl[r][0], sel_copy) : wl[r][0];
      } else {
        const uint32_t bz = (uint32_t)ty[r].z << 16, bw = (uint32_t)ty[r].w << 16;
        pk = 0;
        uint32_t skv[4], h0[4], h1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) skv[k] = fast ? sel[k] : ((uint32_t)k | (0x0cu << 8) | ((uint32_t)(4 + k) << 16) | (0x0cu << 24));
        // The upper source row is the previous output row's lower one four times out of five: its interpolated values were carried
        // over.  need0 is wave-uniform, and the empty asm statement keeps the compiler from turning the block into "compute anyway, then
        // select" (r05: it did - three vector instructions and a v_cndmask per pixel on every row).  The four dot products of a row are
        // issued together and consumed afterwards: v_dot2 needs wait states before its result can be read (the compiler filled them with s_nop).
        if (need0[r]) {
          asm volatile("");
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint32_t p0 = __builtin_amdgcn_perm(wh[r][0], wl[r][0], skv[k]);
            h0[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, p0), __builtin_bit_cast(lv_us2, coef[k]), 0u, false);
          }
#pragma unroll
          for (int k = 0; k < 4; k++) h0[k] >>= 4;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) h0[k] = hc[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t p1 = __builtin_amdgcn_perm(wh[r][1], wl[r][1], skv[k]);
          h1[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, p1), __builtin_bit_cast(lv_us2, coef[k]), 0u, false);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          h1[k] >>= 4;
          hc[k] = h1[k];
          // ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2, weights pre-shifted by 16 for mul_hi
          uint32_t o = (__umulhi(h0[k], bz) + __umulhi(h1[k], bw) + 2u) >> 2;
          // (the value fits a byte; without the opaque copy the compiler rewrites (x >> 2) << 8k as ((x << (8k - 2)) + c) & mask and
          // rematerialises the two literals into vector registers for every pixel: four instructions where add3 / shift / shift-or are three)
          asm("" : "+v"(o));
          pk |= o << (8 * k);
        }
      }
      tile[rr][lane] = pk;
      const int py = Q0 + rr;
      if (own_x && rr >= 3 && rr < 3 + LV_OWN_R && py < PH)
        *reinterpret_cast<uint32_t*>(plane + ((uint32_t)py * (uint32_t)L.stride + (uint32_t)(P0 + 4 * lane))) = pk;
    }
    hc_row = ty[LV_HROWS - 1].y;
    LVP_MARK();
  }
#ifdef LV_STOP      // developer switch (instruction counts): the kernel ends after the resize phase
  return;
#endif
  __syncthreads();
  LVP_MARK();

  // ---- phase 2: GaussianBlur 7x7 of the owned in-image pixels from LDS (fixed point, see orb_blur) ----
  const int tid = threadIdx.x;
  if (tid >= LV_WAVES * 62) return;
  const int c = tid % 62, q = tid / 62;
  const int x0 = P0 + 4 * c + 3 - PS_EDGE;                 // level column of the blur group (multiple of 4)
  if (x0 < 0 || x0 >= L.w) return;
  const int o0 = q * LV_BLUR_ROWS;                         // first owned row of the chunk
  const int nrows = min(LV_BLUR_ROWS, LV_OWN_R - o0);
  const int y0 = Q0 + 3 + o0 - PS_EDGE;                    // level row of the chunk's first output
  if (y0 >= L.h || y0 + nrows <= 0) return;
  // horizontal 7-tap sums (exact in 16 bits: the taps add up to 257); two consecutive rows are packed into one register so that the
  // vertical pass is four v_dot2_u32_u16 per pixel.  The four pixels of a group read bytes j .. j + 6 of the 12-byte window
  // (a0, a1, a2): instead of shifting the window to every pixel (v_alignbyte x 6) the TAPS are shifted - constants - and every
  // aligned dword that holds one of the pixel's bytes gets its own v_dot4_u32_u8: 2 + 2 + 3 + 3 dot products, nothing else (r05).
  constexpr uint32_t T0 = 18, T1 = 34, T2 = 49, T3 = 55, T4 = 49, T5 = 34, T6 = 18;
#define LV_K4(b0, b1, b2, b3) ((uint32_t)(b0) | ((uint32_t)(b1) << 8) | ((uint32_t)(b2) << 16) | ((uint32_t)(b3) << 24))
  uint32_t E[(LV_BLUR_ROWS + 7) / 2][4];     // E[i][j] = hs(row 2i)[j] | hs(row 2i + 1)[j] << 16
#pragma unroll
  for (int r = 0; r < LV_BLUR_ROWS + 6; r++) {
    const int rr = o0 + r;
    const uint32_t a0 = tile[rr][c], a1 = tile[rr][c + 1], a2 = tile[rr][c + 2];
    uint32_t hs[4];
    hs[0] = __builtin_amdgcn_udot4(a1, LV_K4(T4, T5, T6, 0), __builtin_amdgcn_udot4(a0, LV_K4(T0, T1, T2, T3), 0u, false), false);
    hs[1] = __builtin_amdgcn_udot4(a1, LV_K4(T3, T4, T5, T6), __builtin_amdgcn_udot4(a0, LV_K4(0, T0, T1, T2), 0u, false), false);
    hs[2] = __builtin_amdgcn_udot4(a2, LV_K4(T6, 0, 0, 0), __builtin_amdgcn_udot4(a1, LV_K4(T2, T3, T4, T5), __builtin_amdgcn_udot4(a0, LV_K4(0, 0, T0, T1), 0u, false), false), false);
    hs[3] = __builtin_amdgcn_udot4(a2, LV_K4(T5, T6, 0, 0), __builtin_amdgcn_udot4(a1, LV_K4(T1, T2, T3, T4), __builtin_amdgcn_udot4(a0, LV_K4(0, 0, 0, T0), 0u, false), false), false);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (r & 1) E[r >> 1][j] |= hs[j] << 16; else E[r >> 1][j] = hs[j];
    }
  }
#undef LV_K4
  const uint32_t C01 = 18u | (34u << 16), C23 = 49u | (55u << 16), C45 = 49u | (34u << 16), C6 = 18u;           // even output rows
  const uint32_t D0 = 18u << 16, D12 = 34u | (49u << 16), D34 = 55u | (49u << 16), D56 = 34u | (18u << 16);      // odd output rows
  // (32-bit offset from the image's wave-uniform arena base: one vector add per row; a 64-bit row address was two v_mad_u64_u32)
  uint32_t doff = L.blur_off + (uint32_t)x0 + (uint32_t)y0 * (uint32_t)L.bstride;
#pragma unroll
  for (int o = 0; o < LV_BLUR_ROWS; o++, doff += (uint32_t)L.bstride) {
    const int y = y0 + o;
    if (o >= nrows || y < 0 || y >= L.h) continue;
    const int m = o >> 1;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      v[j] = 32768u;
      if (o & 1) {
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m][j]), __builtin_bit_cast(lv_us2, D0), v[j], false);
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m + 1][j]), __builtin_bit_cast(lv_us2, D12), v[j], false);
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m + 2][j]), __builtin_bit_cast(lv_us2, D34), v[j], false);
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m + 3][j]), __builtin_bit_cast(lv_us2, D56), v[j], false);
      } else {
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m][j]), __builtin_bit_cast(lv_us2, C01), v[j], false);
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m + 1][j]), __builtin_bit_cast(lv_us2, C23), v[j], false);
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m + 2][j]), __builtin_bit_cast(lv_us2, C45), v[j], false);
        v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(lv_us2, E[m + 3][j]), __builtin_bit_cast(lv_us2, C6), v[j], false);
      }
    }
    // the four results are v >> 16 (at most 257), saturated to a byte: the upper halves of two sums side by side (v_perm_b32) are a
    // packed i16 pair that v_sat_pk_u8_i16 turns into two bytes - five instructions per dword where shift / min / shift-or were twelve
    *reinterpret_cast<uint32_t*>(base + doff) = ps_sat_pack4_hi16(v[0], v[1], v[2], v[3]);
  }
  LVP_PRINT();
}

// copyMakeBorder(REFLECT_101) of all levels in one launch: every border dword is recomputed from the interior.
// Work items per level: (19 + 19) full rows and, for each interior row, the dword groups that touch the left /
// right border.  (Interior bytes inside such a group are rewritten with their own value.)
__global__ __launch_bounds__(256) void orb_border(OrbPlan plan, uint8_t* arena) {
  const int img = blockIdx.y;
  int level = 0;
#pragma unroll
  for (int l = 1; l < PS_ORB_MAX_LEVELS; l++)
    if (l < plan.nlevels && (int)blockIdx.x >= plan.lv[l].border_blk_base) level = l;
  const OrbLevel L = plan.lv[level];
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int PW = L.w + 2 * PS_EDGE, PH = L.h + 2 * PS_EDGE;
  const int ngx = (PW + 3) >> 2;                                  // dword groups per row
  const int nleft = (PS_EDGE + 3) >> 2;                           // groups touching the left border: 5
  const int rstart = (PS_EDGE + L.w) >> 2;                        // first group touching the right border
  const int nright = ngx - rstart;
  const int n_band = 2 * PS_EDGE * ngx, n_side = L.h * (nleft + nright);
  const int t = (blockIdx.x - L.border_blk_base) * 256 + threadIdx.x;
  if (t >= n_band + n_side) return;
  int py, gx;
  if (t < n_band) {
    const int rr = t / ngx;
    gx = t - rr * ngx;
    py = rr < PS_EDGE ? rr : (L.h + rr);                          // top rows 0..18, bottom rows h+19..h+37
  } else {
    const int u = t - n_band, per = nleft + nright;
    const int rr = u / per, c = u - rr * per;
    py = PS_EDGE + rr;
    gx = c < nleft ? c : rstart + (c - nleft);
  }
  const int sy = reflect101(py - PS_EDGE, L.h);
  const uint8_t* srow = base + L.plane_off + (size_t)(PS_EDGE + sy) * L.stride + PS_EDGE;
  uint32_t pk = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = min(gx * 4 + k, PW - 1);
    pk |= (uint32_t)srow[reflect101(px - PS_EDGE, L.w)] << (8 * k);
  }
  (void)PH;
  *reinterpret_cast<uint32_t*>(base + L.plane_off + (size_t)py * L.stride + gx * 4) = pk;
}

// ------------------------------------------------------------------------------------------------
// FAST cells: ORBextractor.cc:765-829 + OpenCV 3.4 FAST_t<16>/cornerScore<16> with NMS.
//
// For a pixel let s = max over the 16 nine-pixel arcs of min(|signed diff|) (dark or bright arc).
// The pixel is a corner at threshold t iff s > t and OpenCV's score is s - 1, for ANY t (the score
// does not depend on t once the pixel is a corner).  A keypoint at threshold t is a corner whose
// score is strictly greater than the 8 neighbours' scores, where neighbours outside the cell's
// candidate area [3, w-4] x [3, h-4] score 0 (FAST runs on the cell ROI).  Hence:
//   keypoint_t(p) = s(p) > t  AND  s(p) > s(n) for all in-cell neighbours n,
// the cell uses t = iniThFAST unless that yields no keypoint, then minThFAST (ORBextractor.cc:809-816).
//
// One WAVE per FAST cell, four independent cells per workgroup, no workgroup barriers.
//   A  the cell window (cell + 6) goes to LDS shifted to its own origin (8-byte loads, DPP + v_alignbyte)
//   B  necessary test on the 4 compass points of the ring, 4 horizontally adjacent pixels per lane and step on packed u16.
//      Nine contiguous ring pixels always contain two ADJACENT compass points, so a corner at threshold t has two adjacent
//      compass points both darker than v - t or both brighter than v + t.  "Some adjacent pair is below x" is
//      (N < x or S < x) and (E < x or W < x), i.e. max(min(N, S), min(E, W)) < x: three packed operations per polarity.
//      A pixel that passes as dark goes to the dark list, as bright to the bright list (both: to both).  The lists are
//      unordered (slot-major within a step): two v_mbcnt and one address per (slot, polarity).
//   C  exact score of the listed pixels, TWO per lane: a list entry has one polarity, and with r' = 255 - r, v' = 255 - v
//      for bright entries both polarities are  v' - min over the 16 arcs of (max of the arc's 9 values)  on packed u16 halves.
//      A dark and a bright 9-arc cannot coexist on a 16-ring, so a pixel listed twice is a corner in at most one list.
//   D  strict 3x3 NMS on the LDS score map, threshold fallback, raster-ordered emission into the cell's slots: the
//      kept pixels set bits in a per-row bitmap, a row-count prefix gives every keypoint its raster rank.
// ------------------------------------------------------------------------------------------------
// wave-level phase separator: LDS traffic of one wave is processed in order, the fence only keeps the
// compiler from moving accesses across it (no workgroup barrier anywhere in this kernel)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// inclusive prefix sum over the 64 lanes (Kogge-Stone inside the 16-lane DPP rows, then the two row broadcasts)
__device__ __forceinline__ int wave_scan_inclusive(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

#define FAST_T 256
#ifndef FAST_STOP
#define FAST_STOP 0      // developer switch (instruction counts per phase: the kernel returns after phase A / B / C / D; results are wrong)
#endif
typedef unsigned short fs_us2 __attribute__((ext_vector_type(2)));
typedef short fs_s2 __attribute__((ext_vector_type(2)));

// the corner score of two pixels at once on packed u16 halves from their 16 ring values x and centres cv.  BRIGHT = false: v - (min over
// the 16 nine-pixel arcs of the arc's maximum) - a dark corner's score, or a bright one's on complemented values; BRIGHT = true: (max over
// the arcs of the arc's minimum) - v on the values as they are.  The arc extremes come from running extremes of the ring's two 8-blocks
// (suffix S, prefix P): the arc that starts at i < 8 is the block-0 suffix from i and the block-1 prefix up to i, the arc that starts
// at i + 8 wraps the other way (59 packed operations).
template <bool BRIGHT>
__device__ __forceinline__ fs_s2 fast_arc_score(const fs_us2* x, fs_s2 cv) {
  auto inner = [](fs_us2 a, fs_us2 b) { return BRIGHT ? __builtin_elementwise_min(a, b) : __builtin_elementwise_max(a, b); };
  auto outer = [](fs_us2 a, fs_us2 b) { return BRIGHT ? __builtin_elementwise_max(a, b) : __builtin_elementwise_min(a, b); };
  fs_us2 S0[8], P0[8], S1[8], P1[8];
  S0[7] = x[7]; P0[0] = x[0]; S1[7] = x[15]; P1[0] = x[8];
#pragma unroll
  for (int i = 6; i >= 0; i--) { S0[i] = inner(x[i], S0[i + 1]); S1[i] = inner(x[8 + i], S1[i + 1]); }
#pragma unroll
  for (int i = 1; i < 8; i++) { P0[i] = inner(x[i], P0[i - 1]); P1[i] = inner(x[8 + i], P1[i - 1]); }
  fs_us2 best = outer(inner(S0[0], P1[0]), inner(S1[0], P0[0]));
#pragma unroll
  for (int i = 1; i < 8; i++) best = outer(best, outer(inner(S0[i], P1[i]), inner(S1[i], P0[i])));
  return BRIGHT ? __builtin_bit_cast(fs_s2, best) - cv : cv - __builtin_bit_cast(fs_s2, best);
}

// LDS layout of one cell-wave (bytes; every region 16-byte aligned).  TR = window rows of the plan's tallest cell.
struct FastLds { int smap, bmp, list, cap, total; };
__host__ __device__ __forceinline__ FastLds fast_lds(int TS, int TR, int LCAP) {
  FastLds f;
  int o = (TR * TS + 15) & ~15;
  f.smap = o; o += (TS * (TR - 4) + 15) & ~15;         // score map with a zero ring, (rows + 2) x TS: the window's row stride, so that
                                                       // a list entry (offset of the pixel in the window) addresses both
  f.bmp = o; o += (8 * (TR - 6) + 15) & ~15;           // kept-keypoint bitmap, one u64 per cell row
  // survivor entries: dark from the front, bright from the back.  Sized for half of the cell's pixels (a step can add 512);
  // a cell that would overflow it is scored in several rounds (exact, slower: see the kernel).  With the 32 x 40 cells of
  // 1242 x 375 a wave takes 5120 bytes: 8 workgroups per CU.
  f.cap = ((LCAP / 2 > 640 ? LCAP / 2 : 640) + 7) & ~7;
  f.list = o; o += 2 * f.cap;
  f.total = o;
  return f;
}

// TS: LDS row stride of the window; 1 << LG: 4-pixel groups per cell row (8: cells up to 32 px wide, 16: up to 64);
// MAXROWS: window rows the loader is unrolled for.
#ifndef PS_FAST_WAVES
#define PS_FAST_WAVES 8        // waves per SIMD the register allocation aims at (the LDS of the usual cells admits 8 workgroups per CU)
#endif
template <int TS, int LG, int MAXROWS>
__global__ __launch_bounds__(FAST_T) __attribute__((amdgpu_waves_per_eu(PS_FAST_WAVES, 8))) void orb_fast_cells(OrbPlan plan, uint8_t* arena, FastLds F, int nimg, int bpi,
                                                         const uint8_t* masks, int mask_stride, size_t mask_pitch, const uint32_t* celltab) {
  constexpr int NG = 1 << LG, RPS = 64 >> LG;          // groups per row = loader lanes per row; rows per 64-lane step
  constexpr int NPASS = (MAXROWS + RPS - 1) / RPS;
  extern __shared__ __attribute__((aligned(16))) uint8_t fast_smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform geometry stays on the scalar unit
  int img, lb;
  if (!xcd_image_block(nimg, img, lb)) return;
  const int cell = lb * 4 + wave;
  if (cell >= plan.n_cells) return;
  constexpr int SS = TS;                               // score-map row stride = window row stride
  // (r05: the scalar unit is the busiest unit of this kernel - 0.79 against 0.75 for the vector ALUs, tools/valu_busy.sh - : the LDS
  // layout comes from the host, and the cell's level and grid position from a table, one scalar load, where the level was searched and
  // the cell index divided by the grid width)
  uint8_t* tile = fast_smem + (size_t)wave * F.total;
  uint8_t* smap = tile + F.smap;
  unsigned long long* bmp = reinterpret_cast<unsigned long long*>(tile + F.bmp);
  uint16_t* list = reinterpret_cast<uint16_t*>(tile + F.list);
  const int CAP = F.cap;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  typedef const uint32_t __attribute__((address_space(4))) * celltab_ptr;          // wave-uniform address, written once when the plan was made
  const uint32_t crec = ((celltab_ptr)(uintptr_t)celltab)[cell];
  const int level = (int)(crec & 15u), ci_x = (int)((crec >> 4) & 0x3FFFu), ci_y = (int)(crec >> 18);
  const OrbLevel& L = plan.lv[level];
  const int ci = cell - L.cell_base;
  int32_t* cellcnt = reinterpret_cast<int32_t*>(base + plan.cellcnt_off);
  const int maxBX = L.w - PS_MINB, maxBY = L.h - PS_MINB;
  const int iniX = PS_MINB + ci_x * L.w_cell, iniY = PS_MINB + ci_y * L.h_cell;
  const int maxX = min(iniX + L.w_cell + 6, maxBX), maxY = min(iniY + L.h_cell + 6, maxBY);
  const int ww = maxX - iniX, wh = maxY - iniY, cw = ww - 6, ch = wh - 6;
  if (iniX >= maxBX - 3 || iniY >= maxBY - 3 || cw <= 0 || ch <= 0) {
    if (lane == 0) cellcnt[cell] = 0;
    return;
  }
  const int ry = lane >> LG, g = lane & (NG - 1);
  // ---- A: window -> LDS.  NG lanes per row, 8 bytes per lane, RPS rows per pass; every load is issued before the first
  // LDS store so that the window costs one memory round trip.  The rows land shifted to the window's own origin (the next
  // dword of the row comes from the neighbouring lane of the 16-lane DPP row), so that the later stages read aligned dwords. ----
  {
    const int shift = (PS_EDGE + iniX) & 3;   // plane origins are 256-B aligned, strides multiples of 64
    const uint8_t* ga = base + L.plane_off + (size_t)(PS_EDGE + iniY) * L.stride + (PS_EDGE + iniX - shift);
    const bool wact = 8 * g < shift + ww + 4;            // this lane's 8 bytes hold window bytes (or the dword that follows them)
    const uint8_t* gp = ga + (wact ? 8 * g : 0);
    uint2 tmp[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int y = min(RPS * i + ry, wh - 1);
      tmp[i] = *reinterpret_cast<const uint2*>(gp + (size_t)y * L.stride);
    }
    // the score map (zero ring included) is cleared while the loads are in flight
    for (int t = lane; t < (SS * (ch + 2) + 15) / 16; t += 64) reinterpret_cast<uint4*>(smap)[t] = make_uint4(0, 0, 0, 0);
    const bool wst = 8 * g < TS;
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tmp[i].x, 0x101 /* row_shl:1 */, 0xF, 0xF, true);
      uint2 o;
      o.x = __builtin_amdgcn_alignbyte(tmp[i].y, tmp[i].x, (uint32_t)shift);
      o.y = __builtin_amdgcn_alignbyte(nxt, tmp[i].y, (uint32_t)shift);
      if (wst && RPS * i + ry < wh) *reinterpret_cast<uint2*>(tile + (RPS * i + ry) * TS + 8 * g) = o;
    }
  }
  wave_sync();
#if FAST_STOP == 1
  if (lane == 0) cellcnt[cell] = 0;
  return;
#endif
  int total = 0;
  // Two passes at most: first with iniThFAST - a keypoint at threshold t only competes with neighbours that are corners
  // at t, so when the cell has a keypoint at iniThFAST (the common case) the many weak corners between minThFAST and
  // iniThFAST never need a score.  Only a cell without one repeats the pipeline at minThFAST (ORBextractor.cc:809-816).
  for (int pass = 0; pass < 2; pass++) {
    const int th = pass == 0 ? plan.ini_th : plan.min_th;
    // per-lane packed thresholds: a pixel of the lane's group beyond the cell's width (not a multiple of 4) never passes
    fs_s2 thv[2];
#pragma unroll
    for (int pq = 0; pq < 2; pq++) {
      thv[pq].x = (short)(4 * g + 2 * pq < cw ? th : 0x7FFF);
      thv[pq].y = (short)(4 * g + 2 * pq + 1 < cw ? th : 0x7FFF);
    }
    // ---- B + C.  B fills the survivor region step by step; C scores what is listed.  One round unless the region
    // would overflow (more than half of the cell's pixels listed: noise), then C runs on what is there and B goes on. ----
    int nd = 0, nb = 0, y0 = 0;
    bool flushed = false;
    const uint8_t* rp = tile + ry * TS + 4 * g;          // window row of the lane's N points; the centre row is 3 below
    uint32_t val = (uint32_t)(ry * TS + 4 * g);          // list entry of the lane's first pixel: its offset in the window, y * TS + x
    for (;;) {
      bool full = false;
      for (; y0 < ch; y0 += RPS, rp += RPS * TS, val += RPS * TS) {
        // lanes whose row lies below the cell read rows of the window that exist in LDS but mean nothing: masked out (a ballot
        // inside a lane-dependent branch would make every count that follows lane-dependent)
        const int nl = (ch - y0) << LG;
        const unsigned long long rv = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
        const uint32_t n0 = reinterpret_cast<const uint32_t*>(rp)[0], n1 = reinterpret_cast<const uint32_t*>(rp)[1];
        const uint32_t c0 = reinterpret_cast<const uint32_t*>(rp + 3 * TS)[0], c1 = reinterpret_cast<const uint32_t*>(rp + 3 * TS)[1],
                       c2 = reinterpret_cast<const uint32_t*>(rp + 3 * TS)[2];
        const uint32_t s0 = reinterpret_cast<const uint32_t*>(rp + 6 * TS)[0], s1 = reinterpret_cast<const uint32_t*>(rp + 6 * TS)[1];
        unsigned long long DK[4], BR[4];
#pragma unroll
        for (int pq = 0; pq < 2; pq++) {
          // bytes (b, b + 1) of the 8-byte pair {hi, lo} as packed u16
#define PAIR(hi, lo, b) __builtin_bit_cast(fs_us2, __builtin_amdgcn_perm(hi, lo, (uint32_t)(b) | 0x0c000c00u | ((uint32_t)((b) + 1) << 16)))
          const fs_us2 pN = PAIR(n1, n0, 3 + 2 * pq), pS = PAIR(s1, s0, 3 + 2 * pq), pW = PAIR(c1, c0, 2 * pq);
          const fs_us2 pE = pq == 0 ? PAIR(c1, c0, 6) : PAIR(c2, c1, 4);
          const fs_s2 cv = __builtin_bit_cast(fs_s2, PAIR(c1, c0, 3 + 2 * pq));
#undef PAIR
          const fs_us2 M = __builtin_elementwise_max(__builtin_elementwise_min(pN, pS), __builtin_elementwise_min(pE, pW));
          const fs_us2 m = __builtin_elementwise_min(__builtin_elementwise_max(pN, pS), __builtin_elementwise_max(pE, pW));
          const fs_s2 dd = cv - __builtin_bit_cast(fs_s2, M);         // > th: two adjacent compass points darker than v - th
          const fs_s2 bb = __builtin_bit_cast(fs_s2, m) - cv;         // > th: two adjacent compass points brighter than v + th
          DK[2 * pq] = __builtin_amdgcn_ballot_w64(dd.x > thv[pq].x);
          DK[2 * pq + 1] = __builtin_amdgcn_ballot_w64(dd.y > thv[pq].y);
          BR[2 * pq] = __builtin_amdgcn_ballot_w64(bb.x > thv[pq].x);
          BR[2 * pq + 1] = __builtin_amdgcn_ballot_w64(bb.y > thv[pq].y);
        }
        // (scalar-unit diet, r05: the row mask only where a step is partial - the cell's last one -, and the step's survivor count only
        // where the region could overflow at all: a step adds at most 512 entries)
        if (nl < 64) {
#pragma unroll
          for (int i = 0; i < 4; i++) { DK[i] &= rv; BR[i] &= rv; }
        }
        if (nd + nb + 512 > CAP) {
          int nstep = 0;
#pragma unroll
          for (int i = 0; i < 4; i++) nstep += __popcll(DK[i]) + __popcll(BR[i]);
          if (nd + nb + nstep > CAP) { full = true; break; }   // (512 <= CAP: an empty region always takes a step)
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (DK[i]) {                                   // wave-uniform
            // (the running count goes into the scalar base address: as the v_mbcnt accumulator it would cost a move per slot)
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(DK[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)DK[i], 0u));
            uint16_t* wp = list + nd;
            if (__builtin_amdgcn_inverse_ballot_w64(DK[i])) wp[pos] = (uint16_t)(val + i);
            nd += __popcll(DK[i]);
          }
          if (BR[i]) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(BR[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)BR[i], 0u));
            uint8_t* wp = reinterpret_cast<uint8_t*>(list + (CAP - 1 - nb));
            if (__builtin_amdgcn_inverse_ballot_w64(BR[i])) *reinterpret_cast<uint16_t*>(wp + __mul24((int)pos, -2)) = (uint16_t)(val + i);
            nb += __popcll(BR[i]);
          }
        }
      }
      wave_sync();
#if FAST_STOP == 2
      if (lane == 0) cellcnt[cell] = 0;
      return;
#endif
#if FAST_STOP == 6      // second pass only: return after its compass test
      if (pass == 1) { if (lane == 0) cellcnt[cell] = nd + nb; return; }
#endif
      // ---- C: exact scores, two entries per lane: entry k of the concatenation [dark entries, bright entries] ----
      const int ntot = nd + nb;
      for (int i0 = 0; i0 < ntot; i0 += 128) {
        uint32_t p[2], polm = 0;
        const uint8_t* q[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int kk = min(i0 + 64 * e + lane, ntot - 1);   // lanes beyond the end repeat the last entry (their result is dropped)
          const bool bright = kk >= nd;
          p[e] = list[bright ? CAP - 1 + nd - kk : kk];
          if (bright) polm |= 0xFFu << (16 * e);
          q[e] = tile + p[e];                                  // window position of the pixel's 7 x 7 neighbourhood
        }
        // ring offsets from q (centre at (3, 3)): position 0 is (dx, dy) = (0, +3), then as OpenCV's table
        constexpr int RO[16] = {6 * TS + 3, 6 * TS + 4, 5 * TS + 5, 4 * TS + 6, 3 * TS + 6, 2 * TS + 6, 1 * TS + 5, 0 * TS + 4,
                                0 * TS + 3, 0 * TS + 2, 1 * TS + 1, 2 * TS + 0, 3 * TS + 0, 4 * TS + 0, 5 * TS + 1, 6 * TS + 2};
        // r05: a round whose 128 entries are all dark (or all bright) - wave-uniform - needs no per-lane complement: dark rounds take the
        // ring as it is, bright rounds run the same recurrences with minimum and maximum exchanged; only the round that holds the
        // boundary between the two lists complements per lane (the exclusive-or was 18 of a round's ~120 vector instructions)
        const int klast = min(i0 + 127, ntot - 1);
        fs_us2 x[16];
        fs_s2 sc;
        if (klast < nd) {
#pragma unroll
          for (int i = 0; i < 16; i++) x[i] = __builtin_bit_cast(fs_us2, (uint32_t)q[0][RO[i]] | ((uint32_t)q[1][RO[i]] << 16));
          sc = fast_arc_score<false>(x, __builtin_bit_cast(fs_s2, (uint32_t)q[0][3 * TS + 3] | ((uint32_t)q[1][3 * TS + 3] << 16)));
        } else if (i0 >= nd) {
#pragma unroll
          for (int i = 0; i < 16; i++) x[i] = __builtin_bit_cast(fs_us2, (uint32_t)q[0][RO[i]] | ((uint32_t)q[1][RO[i]] << 16));
          sc = fast_arc_score<true>(x, __builtin_bit_cast(fs_s2, (uint32_t)q[0][3 * TS + 3] | ((uint32_t)q[1][3 * TS + 3] << 16)));
        } else {
#pragma unroll
          for (int i = 0; i < 16; i++)
            x[i] = __builtin_bit_cast(fs_us2, ((uint32_t)q[0][RO[i]] | ((uint32_t)q[1][RO[i]] << 16)) ^ polm);
          sc = fast_arc_score<false>(x, __builtin_bit_cast(fs_s2, ((uint32_t)q[0][3 * TS + 3] | ((uint32_t)q[1][3 * TS + 3] << 16)) ^ polm));
        }
        if ((int)sc.x > th && i0 + lane < ntot) smap[p[0] + (SS + 1)] = (uint8_t)sc.x;
        if ((int)sc.y > th && i0 + 64 + lane < ntot) smap[p[1] + (SS + 1)] = (uint8_t)sc.y;
      }
#if FAST_STOP == 3
      if (lane == 0) cellcnt[cell] = 0;
      return;
#endif
#if FAST_STOP == 7      // second pass only: return after its scores
      if (pass == 1) { if (lane == 0) cellcnt[cell] = 0; return; }
#endif
      if (!full) break;
      wave_sync();
      nd = nb = 0;
      flushed = true;
    }
    wave_sync();
    // ---- D: strict 3x3 NMS of the corners at th (the non-zero entries of the score map: scores do not depend on th, and
    // only scores above the current or an earlier, higher th were written) ----
    unsigned long long anykp = 0;
    unsigned long long rowbits = 0;                      // several rounds: lane = cell row
    if (!flushed) {
      const int ntot = nd + nb;
      for (int i0 = 0; i0 < ntot; i0 += 64) {
        const int k = i0 + lane;
        bool lmax = false;
        if (k < ntot) {
          const int li = k >= nd ? CAP - 1 + nd - k : k;
          const int pp = list[li];
          const uint8_t* m = smap + pp + (SS + 1);
          const int s = m[0];
          lmax = s > m[-1] && s > m[1] && s > m[-SS - 1] && s > m[-SS] && s > m[-SS + 1] && s > m[SS - 1] && s > m[SS] && s > m[SS + 1];
          if (lmax) list[li] = (uint16_t)(pp | 0x8000); // bit 15: survives the NMS (a pixel listed twice is marked twice)
        }
        anykp |= __builtin_amdgcn_ballot_w64(lmax);
      }
    } else if (lane < ch) {
      for (int x = 0; x < cw; x++) {
        const uint8_t* m = smap + (lane + 1) * SS + x + 1;
        const int s = m[0];
        if (s > m[-1] && s > m[1] && s > m[-SS - 1] && s > m[-SS] && s > m[-SS + 1] && s > m[SS - 1] && s > m[SS] && s > m[SS + 1])
          rowbits |= 1ull << x;
      }
    }
    if (flushed) anykp = __builtin_amdgcn_ballot_w64(rowbits != 0);
#if FAST_STOP == 4
    if (lane == 0) cellcnt[cell] = 0;
    return;
#endif
#if FAST_STOP == 5
    if (!anykp && pass == 0) { if (lane == 0) cellcnt[cell] = 0; return; }
#endif
    if (!anykp && pass == 0) continue;   // no keypoint at iniThFAST: run the cell again at minThFAST
    // ---- emission in raster order: bitmap of the kept pixels, one row per lane; the row counts' prefix is the rank ----
    const int mx0 = ci_x * L.w_cell + 3, my0 = ci_y * L.h_cell + 3;   // cell pixel (0, 0) relative to (minBorderX, minBorderY)
    if (!flushed) {
      if (lane < ch) bmp[lane] = 0;
      wave_sync();
      const int ntot = nd + nb;
      for (int i0 = 0; i0 < ntot; i0 += 64) {
        const int k = i0 + lane;
        const int pv = k < ntot ? list[k >= nd ? CAP - 1 + nd - k : k] : 0;
        if (pv & 0x8000) { const int po = pv & 0x7FFF, py = po / TS; atomicOr(&bmp[py], 1ull << (po - py * TS)); }
      }
      wave_sync();
      rowbits = lane < ch ? bmp[lane] : 0ull;
    }
    if (masks) {
      // object-feature variant (SURVEY.md 8f-2 stand-in): a keypoint whose pixel in the level-0 image - cvRound(level
      // coordinate * scale), clipped - lies outside the mask never reaches the quadtree
      unsigned long long w = rowbits;
      while (w) {
        const int x = __ffsll((long long)w) - 1;
        w &= w - 1;
        const int px = x + mx0 + PS_MINB, py = lane + my0 + PS_MINB;
        const int mx = min(max(__float2int_rn(__fmul_rn((float)px, L.scale)), 0), plan.img_w - 1);
        const int my = min(max(__float2int_rn(__fmul_rn((float)py, L.scale)), 0), plan.img_h - 1);
        if (masks[(size_t)img * mask_pitch + (size_t)my * mask_stride + mx] == 0) rowbits &= ~(1ull << x);
      }
    }
    {
      const int c = __popcll(rowbits);
      const int inc = wave_scan_inclusive(c);
      total = __builtin_amdgcn_readlane(inc, 63);
      int pos = inc - c;
      uint32_t* slots = reinterpret_cast<uint32_t*>(base + plan.cand_base) + L.cand_off + (size_t)ci * L.cell_cap;
      while (rowbits) {
        const int x = __ffsll((long long)rowbits) - 1;
        rowbits &= rowbits - 1;
        const int s = smap[(lane + 1) * SS + x + 1];
        // coordinates relative to (minBorderX, minBorderY): local + j*wCell (ORBextractor.cc:822-824)
        if (pos < L.cell_cap) slots[pos] = (uint32_t)(x + mx0) | ((uint32_t)(lane + my0) << 12) | ((uint32_t)s << 24);
        pos++;
      }
    }
    break;
  }   // pass
  if (lane == 0) cellcnt[cell] = min(total, L.cell_cap);
}

// ------------------------------------------------------------------------------------------------
// Quadtree distribution: ORBextractor.cc:539-763 (DistributeOctTree) + :481-537 (DivideNode).
//
// The reference keeps a std::list of nodes, each owning a vector of keys.  Here a key only carries
// its node index; the list is an array in list order that is rebuilt after every pass:
//   new list = [children of the processed parents, last processed parent first, each parent's
//               children in order n4,n3,n2,n1 (they were push_front-ed n1..n4)] ++ [unprocessed
//               nodes in their old order].
// A "full" pass (the outer while) processes every node with more than one key in list order; a
// "careful" round (the inner while, entered when size + 3*nToExpand > N) processes the nodes created
// in the previous pass in descending (key count, address) order and stops as soon as size >= N.
// The heap-address tie-break of the reference is modelled by node creation order.
// Key order inside a node only matters for the final "first maximum wins" selection, which is
// reproduced by taking max (score, -original index).
// ------------------------------------------------------------------------------------------------
#ifndef QT_T
#define QT_T 512
#endif
#define QT_INV 0x80000000u
#ifndef QT_GB
#define QT_GB 4                  // keys per thread and step of the gather (r05: 8 / 12 keys - two / one step for a level-0 workgroup's 4 400 keys - made the
                                  // whole kernel slower, 0.089 -> 0.098 ms per 128 images: registers)
#endif

#ifndef QT_KCAP512
#define QT_KCAP512 2048          // ... of the 512-node configuration
#endif
#ifndef QT_KCAP
#define QT_KCAP 2048              // keys kept in LDS (levels with more candidates use the global scratch); sized so that three
                                  // workgroups share a CU: the kernel is latency-bound and its duration is rounds x workgroup latency
#endif
template <int NCAP, int NT, int KCAP>
struct __attribute__((aligned(16))) QtShared {     // (16: `ord` is read as uint4 in the careful rounds)
  uint32_t boxa[2][NCAP];   // x0 | y0 << 16
  uint32_t boxb[2][NCAP];   // x1 | y1 << 16
  uint32_t cnt[2][NCAP];    // key count | QT_INV (member of vSizeAndPointerToNode)
  uint32_t seq[2][NCAP];    // creation order
  uint32_t child[NCAP * 4];
  int32_t rank[NCAP];       // node -> rank in processing order, -1 = not a candidate
  int32_t ord[NCAP];        // rank -> node
  int32_t cpre[NCAP + 1];   // exclusive prefix (rank order) of non-empty child counts
  uint32_t kxy[KCAP];    // key position x | y << 16
  uint32_t kns[KCAP];    // key node | score << 16
  int32_t surv[NCAP + 1];   // node -> new index when it survives unprocessed (also: cell offsets during the gather)
  int32_t tmp[NT];
  int32_t total;
  int32_t cut;
  int32_t n_expand;
};

// exclusive in-place scan of a[0..n) (n <= capacity of a), returns the total to every thread.
template <int NCAP, int NT, int KCAP>
__device__ int qt_exscan(int32_t* a, int n, QtShared<NCAP, NT, KCAP>& s) {
  const int t = threadIdx.x;
  const int chunk = (n + NT - 1) / NT;
  const int b = min(t * chunk, n), e = min(b + chunk, n);
  int sum = 0;
  for (int i = b; i < e; i++) sum += a[i];
  s.tmp[t] = sum;
  __syncthreads();
  if (t < 64) {
    int loc[NT / 64];
    int ss = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) { loc[k] = ss; ss += s.tmp[t * (NT / 64) + k]; }
    int incl = ss;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (t >= d) incl += o;
    }
    const int excl = incl - ss;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) s.tmp[t * (NT / 64) + k] = excl + loc[k];
    if (t == 63) s.total = incl;
  }
  __syncthreads();
  int run = s.tmp[t];
  for (int i = b; i < e; i++) { const int v = a[i]; a[i] = run; run += v; }
  const int total = s.total;
  __syncthreads();
  return total;
}

__device__ __forceinline__ int qt_quadrant(uint32_t kxy, uint32_t ba, uint32_t bb) {
  const int x0 = ba & 0xFFFF, y0 = ba >> 16, x1 = bb & 0xFFFF, y1 = bb >> 16;
  const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);  // ceil(float(d)/2)
  const int kx = kxy & 0xFFFF, ky = kxy >> 16;
  return (kx < mx ? 0 : 1) + (ky < my ? 0 : 2);
}

#ifdef PS_QT_PROFILE   // developer build: 100 MHz ticks per phase of (level 0, image 0), printed by the kernel
#define QTP_DECL long long qt_t0 = wall_clock64(), qt_tt = qt_t0, qt_ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int qt_np = 0
#define QTP_MARK(k) do { const long long _n = wall_clock64(); qt_ph[k] += _n - qt_tt; qt_tt = _n; } while (0)
#define QTP_PASS() (qt_np++)
#define QTP_PRINT() do { if (threadIdx.x == 0 && blockIdx.x == 0) printf("qt n=%d passes %d ticks: gather %lld A %lld B %lld C %lld cpre %lld E %lld F %lld init %lld final %lld total %lld\n", n, qt_np, qt_ph[0], qt_ph[1], qt_ph[2], qt_ph[3], qt_ph[4], qt_ph[5], qt_ph[6], qt_ph[7], qt_ph[8], wall_clock64() - qt_t0); } while (0)
#else
#define QTP_DECL
#define QTP_MARK(k)
#define QTP_PASS()
#define QTP_PRINT()
#endif
// NT threads per workgroup and KCAP keys in LDS: 512 / 2048 for the large levels; the small ones (node capacity 256) take 256 / 1024,
// half the LDS and twice the workgroups per CU - the kernel is bound by the latency of its passes, not by work.
template <int NCAP, int NT, int KCAP>
__global__ __launch_bounds__(NT) void orb_quadtree(OrbPlan plan, uint8_t* arena, int nimg, int level0) {
  __shared__ QtShared<NCAP, NT, KCAP> s;
  QTP_DECL;
  // level-major block order: the long-running workgroups (level 0, the largest quota) are dispatched first
  const int lrel = blockIdx.x / nimg, img = blockIdx.x - lrel * nimg, t = threadIdx.x, level = level0 + lrel;
  const OrbLevel L = plan.lv[level];
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int32_t* cellcnt = reinterpret_cast<const int32_t*>(base + plan.cellcnt_off) + L.cell_base;
  const uint32_t* slots = reinterpret_cast<const uint32_t*>(base + plan.cand_base) + L.cand_off;
  uint32_t* gkxy = reinterpret_cast<uint32_t*>(base + plan.key_base) + L.key_off;   // global scratch (large levels)
  uint32_t* sel = reinterpret_cast<uint32_t*>(base + plan.sel_base) + L.sel_off;
  int32_t* selcnt = reinterpret_cast<int32_t*>(base + plan.selcnt_off);
  int32_t* ncand_out = reinterpret_cast<int32_t*>(base + plan.ncand_off);
  const int N = L.quota;
  const int ncell = L.n_cols * L.n_rows;

  // ---- gather the cells' candidates in reference emission order (cell-major, raster inside) ----
  // the cell offsets borrow rank / ord / cpre (contiguous, 3 * NCAP + 1 entries, not yet in use), so the cell count of a level
  // is not tied to the node capacity
  int32_t* coff = s.rank;
  for (int c = t; c < ncell; c += NT) coff[c] = cellcnt[c];
  for (int i = t; i < L.n_ini * 4; i += NT) s.child[i] = 0;
  __syncthreads();
  const int n = qt_exscan(coff, ncell, s);
  if (t == 0) ncand_out[level] = n;
  if (n == 0) {
    if (t == 0) selcnt[level] = 0;
    return;
  }
  // flattened gather: key j lives in cell c with off[c] <= j < off[c+1] (binary search in LDS), so every thread
  // issues independent slot loads instead of walking the cells one dependent global round trip at a time
  // keys live in LDS when they fit, else in the level's global scratch: the passes below are compiled once for either address
  // space (a generic pointer would make every key access a FLAT instruction, which takes the slow path to LDS)
  auto passes = [&](auto kxy, auto kns) {
  if (t == 0) coff[ncell] = n;
  __syncthreads();
  for (int j0 = t; j0 < n; j0 += QT_GB * NT) {       // QT_GB keys per thread and step: their slot loads are in flight together
    uint32_t ev[QT_GB];
#pragma unroll
    for (int u = 0; u < QT_GB; u++) {
      const int j = min(j0 + u * NT, n - 1);
      int lo = 0, hi = ncell;          // invariant: off[lo] <= j < off[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (coff[mid] <= j) lo = mid; else hi = mid;
      }
      ev[u] = slots[(size_t)lo * L.cell_cap + (j - coff[lo])];
    }
#pragma unroll
    for (int u = 0; u < QT_GB; u++) {
      const int j = j0 + u * NT;
      if (j >= n) continue;
      const uint32_t e = ev[u];
      const uint32_t x = e & 0xFFF, y = (e >> 12) & 0xFFF, sc = e >> 24;
      // vpIniNodes[kp.pt.x / hX] (ORBextractor.cc:569): float division, truncation
      const int ni = (int)__fdiv_rn((float)x, L.h_x);
      kxy[j] = x | (y << 16);
      kns[j] = (uint32_t)ni | (sc << 16);
      atomicAdd(&s.child[ni], 1u);
    }
  }
  __syncthreads();
  QTP_MARK(0);
  // ---- initial nodes (ORBextractor.cc:543-586); empty ones stay in the array with count 0 and
  // are dropped at the first rebuild, which is when the reference has already erased them --------
  int cur = 0;
  for (int i = t; i < L.n_ini; i += NT) {
    const int x0 = (int)__fmul_rn(L.h_x, (float)i), x1 = (int)__fmul_rn(L.h_x, (float)(i + 1));
    s.boxa[0][i] = (uint32_t)x0;                               // y0 = 0
    s.boxb[0][i] = (uint32_t)x1 | ((uint32_t)(L.h - 2 * PS_MINB) << 16);
    s.cnt[0][i] = s.child[i];
    s.seq[0][i] = (uint32_t)i;
  }
  __syncthreads();
  int A = L.n_ini;   // array length
  int nn = 0;        // list size = nodes with keys
  for (int i = 0; i < L.n_ini; i++) nn += s.cnt[0][i] > 0 ? 1 : 0;
  uint32_t seq_base = (uint32_t)L.n_ini;
  bool careful = false;

  for (;;) {
    const int prev_size = nn;
    uint32_t* boxa = s.boxa[cur]; uint32_t* boxb = s.boxb[cur];
    uint32_t* cnt = s.cnt[cur]; uint32_t* sq = s.seq[cur];
    // A: clear child counters
    for (int i = t; i < A * 4; i += NT) s.child[i] = 0;
    if (t == 0) { s.cut = 0x7fffffff; s.n_expand = 0; }
    __syncthreads();
    QTP_MARK(1);
    // B: count keys per child of every candidate node
    // (four keys per thread and step, their words requested together: the keys of the large levels live in global memory - r05:
    // one dependent round trip per key and pass before)
    for (int k0 = t; k0 < n; k0 += 4 * NT) {
      uint32_t kv[4], kp[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = min(k0 + u * NT, n - 1); kv[u] = kns[k]; kp[u] = kxy[k]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k0 + u * NT >= n) continue;
        const int i = kv[u] & 0xFFFF;
        const uint32_t c = cnt[i];
        const bool cand = careful ? (c & QT_INV) != 0 : (c & ~QT_INV) > 1;
        if (cand) atomicAdd(&s.child[i * 4 + qt_quadrant(kp[u], boxa[i], boxb[i])], 1u);
      }
    }
    __syncthreads();
    QTP_MARK(2);
    // C: processing order
    int ncand;
    if (!careful) {
      for (int i = t; i < A; i += NT) s.rank[i] = (cnt[i] & ~QT_INV) > 1 ? 1 : 0;
      __syncthreads();
      // exclusive scan -> rank; remember candidacy in ord[] temporarily
      for (int i = t; i < A; i += NT) s.ord[i] = s.rank[i];
      __syncthreads();
      ncand = qt_exscan(s.rank, A, s);
      for (int i = t; i < A; i += NT) if (!s.ord[i]) s.rank[i] = -1;
      __syncthreads();
    } else {
      // descending (count, seq): rank = number of candidates that sort after this one
      int local = 0;
      if (n < 65535 && seq_base < 65536u) {
        // (r05) the pair as ONE word, count << 16 | creation order (+ 1; 0 = not a candidate), in `ord` - free until the ranks are known -
        // and four of them per LDS read: the comparison of two fields per node and an LDS word per field was a fifth of a level-0
        // workgroup's time (-DPS_QT_PROFILE)
        uint32_t* key = reinterpret_cast<uint32_t*>(s.ord);
        const int A4 = (A + 3) & ~3;
        for (int i = t; i < A4; i += NT) {
          const uint32_t ci = i < A ? cnt[i] : 0u;
          key[i] = (ci & QT_INV) ? (((ci & ~QT_INV) << 16) | sq[i]) + 1u : 0u;
        }
        __syncthreads();
        for (int i = t; i < A; i += NT) {
          int r = -1;
          const uint32_t ki = key[i];
          if (ki) {
            r = 0;
            for (int j = 0; j < A4; j += 4) {
              const uint4 k = *reinterpret_cast<const uint4*>(key + j);
              r += (k.x > ki ? 1 : 0) + (k.y > ki ? 1 : 0) + (k.z > ki ? 1 : 0) + (k.w > ki ? 1 : 0);
            }
            local++;
          }
          s.rank[i] = r;
        }
      } else
      for (int i = t; i < A; i += NT) {
        int r = -1;
        const uint32_t ci = cnt[i];
        if (ci & QT_INV) {
          r = 0;
          const uint32_t cc = ci & ~QT_INV, si = sq[i];
          for (int j = 0; j < A; j++) {
            const uint32_t cj = cnt[j];
            if (cj & QT_INV) {
              const uint32_t cjj = cj & ~QT_INV;
              r += (cjj > cc || (cjj == cc && sq[j] > si)) ? 1 : 0;
            }
          }
          local++;
        }
        s.rank[i] = r;
      }
      // number of candidates: wave sums, then the NT / 64 partial sums by every thread (r03: one thread adding NT LDS words was a tenth
      // of a level-0 workgroup's time)
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
      if ((t & 63) == 0) s.tmp[t >> 6] = local;
      __syncthreads();
      ncand = 0;
#pragma unroll
      for (int w = 0; w < NT / 64; w++) ncand += s.tmp[w];
      __syncthreads();
    }
    for (int i = t; i < A; i += NT) if (s.rank[i] >= 0) s.ord[s.rank[i]] = i;
    __syncthreads();
    QTP_MARK(3);
    // non-empty children per candidate, in rank order
    for (int r = t; r < ncand; r += NT) {
      const int i = s.ord[r];
      s.cpre[r] = (s.child[i * 4] > 0) + (s.child[i * 4 + 1] > 0) + (s.child[i * 4 + 2] > 0) +
                  (s.child[i * 4 + 3] > 0);
    }
    if (t == 0) s.cpre[ncand] = 0;
    __syncthreads();
    qt_exscan(s.cpre, ncand + 1, s);   // cpre[r] = sum_{r'<r}, cpre[ncand] = total
    // cutoff: first rank after which size >= N (careful rounds only, ORBextractor.cc:729-730)
    int m = ncand - 1;
    if (careful) {
      for (int r = t; r < ncand; r += NT)
        if (nn + s.cpre[r + 1] - (r + 1) >= N) atomicMin(&s.cut, r);
      __syncthreads();
      if (s.cut != 0x7fffffff) m = s.cut;
    }
    const int nproc = m + 1;
    const int TC = nproc > 0 ? s.cpre[nproc] : 0;
    // survivors: nodes with keys that are not processed
    for (int i = t; i < A; i += NT) {
      const bool processed = s.rank[i] >= 0 && s.rank[i] <= m;
      s.surv[i] = (!processed && (cnt[i] & ~QT_INV) > 0) ? 1 : 0;
    }
    __syncthreads();
    const int nsurv = qt_exscan(s.surv, A, s);
    const int nn_new = TC + nsurv;
    QTP_MARK(4);
    // E: write the new list into the other buffer
    uint32_t* nboxa = s.boxa[cur ^ 1]; uint32_t* nboxb = s.boxb[cur ^ 1];
    uint32_t* ncnt = s.cnt[cur ^ 1]; uint32_t* nsq = s.seq[cur ^ 1];
    for (int r = t; r < nproc; r += NT) {
      const int i = s.ord[r];
      const uint32_t ba = boxa[i], bb = boxb[i];
      const int x0 = ba & 0xFFFF, y0 = ba >> 16, x1 = bb & 0xFFFF, y1 = bb >> 16;
      const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
      const uint32_t c0 = s.child[i * 4], c1 = s.child[i * 4 + 1], c2 = s.child[i * 4 + 2], c3 = s.child[i * 4 + 3];
      const uint32_t cc[4] = {c0, c1, c2, c3};
      const uint32_t ca[4] = {(uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)mx | ((uint32_t)y0 << 16),
                              (uint32_t)x0 | ((uint32_t)my << 16), (uint32_t)mx | ((uint32_t)my << 16)};
      const uint32_t cb[4] = {(uint32_t)mx | ((uint32_t)my << 16), (uint32_t)x1 | ((uint32_t)my << 16),
                              (uint32_t)mx | ((uint32_t)y1 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16)};
      const int cr = s.cpre[r + 1] - s.cpre[r];
      const int pbase = TC - s.cpre[r] - cr;   // children of later-processed parents come first
      int before = 0;                           // non-empty children with smaller q
      int expand = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (cc[q] > 0) {
          const int after = cr - before - 1;    // non-empty children with larger q precede it
          const int pos = pbase + after;
          nboxa[pos] = ca[q];
          nboxb[pos] = cb[q];
          ncnt[pos] = cc[q] | (cc[q] > 1 ? QT_INV : 0u);
          nsq[pos] = seq_base + (uint32_t)(s.cpre[r] + before);
          expand += cc[q] > 1 ? 1 : 0;
          before++;
        }
      }
      if (expand) atomicAdd(&s.n_expand, expand);
    }
    for (int i = t; i < A; i += NT) {
      const bool processed = s.rank[i] >= 0 && s.rank[i] <= m;
      if (!processed && (cnt[i] & ~QT_INV) > 0) {
        const int pos = TC + s.surv[i];
        nboxa[pos] = boxa[i];
        nboxb[pos] = boxb[i];
        ncnt[pos] = cnt[i] & ~QT_INV;
        nsq[pos] = sq[i];
      }
    }
    QTP_MARK(5);
    // F: re-home the keys
    for (int k0 = t; k0 < n; k0 += 4 * NT) {
      uint32_t kv4[4], kp4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = min(k0 + u * NT, n - 1); kv4[u] = kns[k]; kp4[u] = kxy[k]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * NT;
        if (k >= n) continue;
        const uint32_t kv = kv4[u];
        const int i = kv & 0xFFFF;
        const int r = s.rank[i];
        int ni;
        if (r >= 0 && r <= m) {
          const int q = qt_quadrant(kp4[u], boxa[i], boxb[i]);
          const int cr = s.cpre[r + 1] - s.cpre[r];
          int before = 0;
          for (int qq = 0; qq < q; qq++) before += s.child[i * 4 + qq] > 0 ? 1 : 0;
          ni = TC - s.cpre[r] - cr + (cr - before - 1);
        } else {
          ni = TC + s.surv[i];
        }
        kns[k] = (kv & 0xFFFF0000u) | (uint32_t)ni;
      }
    }
    __syncthreads();
    const int n_expand = s.n_expand;
    __syncthreads();
    QTP_MARK(6);
    QTP_PASS();
    seq_base += (uint32_t)TC;
    A = nn_new;
    nn = nn_new;
    cur ^= 1;
    // termination: ORBextractor.cc:669-737
    if (nn >= N || nn == prev_size) break;
    if (!careful && nn + 3 * n_expand > N) careful = true;
  }

  QTP_MARK(7);
  // ---- retain the best key of every node (ORBextractor.cc:742-760): max response, first wins ----
  uint32_t* best = s.child;
  for (int i = t; i < nn; i += NT) best[i] = 0;
  __syncthreads();
  for (int k = t; k < n; k += NT) {
    const uint32_t kv = kns[k];
    atomicMax(&best[kv & 0xFFFF], ((kv >> 16) << 20) | (0xFFFFFu - (uint32_t)k));
  }
  __syncthreads();
  for (int i = t; i < nn; i += NT) {
    const uint32_t bv = best[i];
    const uint32_t k = 0xFFFFFu - (bv & 0xFFFFFu);
    const uint32_t xy = kxy[k];
    const uint32_t x = (xy & 0xFFFF) + PS_MINB, y = (xy >> 16) + PS_MINB;   // ORBextractor.cc:843-844
    if (i < L.sel_cap) sel[i] = x | (y << 12) | ((bv >> 20) << 24);
  }
  if (t == 0) selcnt[level] = min(nn, L.sel_cap);
  QTP_MARK(8);
  };
  typedef __attribute__((address_space(3))) uint32_t* lds_keys;
  typedef __attribute__((address_space(1))) uint32_t* glb_keys;
  if (n <= KCAP) passes((lds_keys)s.kxy, (lds_keys)s.kns);
  else passes((glb_keys)gkxy, (glb_keys)(gkxy + L.key_cap));      // node | score << 16
  QTP_PRINT();
}

// ------------------------------------------------------------------------------------------------
// GaussianBlur(7x7, sigma 2, REFLECT_101) on CV_8U as OpenCV 3.4.3 computes it (fixed-point path):
// 8.8 kernel {18,34,49,55,49,34,18}, horizontal sums exact in 16 bits, vertical in 32 bits, one
// rounding (x + 2^15) >> 16, saturated.  ORBextractor.cc:1085-1086.  The padded plane's border is
// REFLECT_101 of the level, so reading the padded plane reproduces the border handling.
// One wave = a strip of 64 columns; each lane slides a 7-deep window of horizontal sums down ROWS.
// ------------------------------------------------------------------------------------------------
// One launch for all levels.  A lane owns 4 adjacent columns (3 aligned dword loads per row: pixel x0-3 sits at
// byte 16 + x0 of the padded row because the border is 19) and BLUR_ROWS output rows; all BLUR_ROWS + 6 row loads
// are issued before any arithmetic (memory-level parallelism), the 7x7 is evaluated separably in registers.
#define BLUR_ROWS 8
__global__ __launch_bounds__(256) void orb_blur(OrbPlan plan, uint8_t* arena) {
  const int img = blockIdx.y;
  int level = 0;
#pragma unroll
  for (int l = 1; l < PS_ORB_MAX_LEVELS; l++)
    if (l < plan.nlevels && (int)blockIdx.x >= plan.lv[l].blur_blk_base) level = l;
  const OrbLevel L = plan.lv[level];
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int nsx = (L.w + 255) >> 8;
  const int bidx = blockIdx.x - L.blur_blk_base;
  const int gy = bidx / nsx, sxi = bidx - gy * nsx;
  const int x0 = sxi * 256 + (threadIdx.x & 63) * 4;
  const int y0 = (gy * 4 + (threadIdx.x >> 6)) * BLUR_ROWS;
  if (x0 >= L.w || y0 >= L.h) return;
  const uint8_t* src = base + L.plane_off + (size_t)PS_EDGE * L.stride + 16 + x0;   // pixel (x0 - 3, 0)
  uint32_t a[BLUR_ROWS + 6][3];
#pragma unroll
  for (int r = 0; r < BLUR_ROWS + 6; r++) {
    const int y = min(y0 + r - 3, L.h + 2);   // rows past the level's bottom border are never used
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (ptrdiff_t)y * L.stride);
    a[r][0] = p[0]; a[r][1] = p[1]; a[r][2] = p[2];
  }
  uint32_t hs[BLUR_ROWS + 6][4];
#pragma unroll
  for (int r = 0; r < BLUR_ROWS + 6; r++) {
#define BB(i) ((a[r][(i) >> 2] >> (((i) & 3) * 8)) & 0xFFu)
#pragma unroll
    for (int j = 0; j < 4; j++)
      hs[r][j] = 18u * (BB(j) + BB(j + 6)) + 34u * (BB(j + 1) + BB(j + 5)) + 49u * (BB(j + 2) + BB(j + 4)) + 55u * BB(j + 3);
#undef BB
  }
  uint8_t* dst = base + L.blur_off + x0;
#pragma unroll
  for (int o = 0; o < BLUR_ROWS; o++) {
    if (y0 + o >= L.h) break;
    uint32_t pk = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t v = 18u * (hs[o][j] + hs[o + 6][j]) + 34u * (hs[o + 1][j] + hs[o + 5][j]) + 49u * (hs[o + 2][j] + hs[o + 4][j]) + 55u * hs[o + 3][j];
      const uint32_t q = (v + 32768u) >> 16;
      pk |= (q > 255u ? 255u : q) << (8 * j);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)(y0 + o) * L.bstride) = pk;
  }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor: ORBextractor.cc:77-104 (IC_Angle, cv::fastAtan2), :108-147
// (computeOrbDescriptor), :1095-1101 (scaling), one wave per selected keypoint.
// ------------------------------------------------------------------------------------------------
#include "describe_common.h"

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  // OpenCV 3.4 atan_f32 (mathfuncs_core.simd.hpp), evaluated without FMA contraction
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

struct PsKeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };
// FOUR keypoints per wave, one per 16-lane DPP row: the angle (atan2, double-precision sincos) is the same ~100 instructions
// whether 16 or 64 lanes share a keypoint, and the reductions stay inside a DPP row.  Levels start at multiples of four
// slots (orb_host.hip), so a wave's four slots belong to one level and the level geometry stays on the scalar unit.
#ifndef PS_DESC_WAVES
#define PS_DESC_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PS_DESC_WAVES, 8))) void orb_describe(OrbPlan plan, uint8_t* arena, PsKeyPoint* out_kps,
                                                   uint8_t* out_desc, int32_t* out_counts, int nimg, int bpi) {
  // the two lane-indexed tables go to LDS once per workgroup (from the vector cache they would be two thirds of the bytes a wave loads)
  __shared__ uint4 s_pat[4][16];
  __shared__ uint2 s_icm[8][16];
  int img, lb;
  if (!xcd_image_block(nimg, img, lb)) return;         // (the whole workgroup)
  // Everything the keypoint's patch address depends on is requested HERE, in front of the barrier, in one go: the tables' words,
  // the wave's slot and the eight per-level counts (two scalar loads).  r05: the counts were read level by level inside
  // `if (l < nlevels)` - eight scalar round trips one after the other, behind the tables' round trip and the barrier, in front of
  // the slot's - which is why the kernel issued on half of its cycles.
  const uint32_t t_pat = reinterpret_cast<const uint32_t*>(&c_pattab)[threadIdx.x], t_icm = reinterpret_cast<const uint32_t*>(&c_ictab)[threadIdx.x];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot0 = (lb * 4 + wv) * 4;
  const int lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  int level = 0;
#pragma unroll
  for (int l = 1; l < PS_ORB_MAX_LEVELS; l++)
    if (l < plan.nlevels && slot0 >= plan.lv[l].sel_off) level = l;
  const OrbLevel& L = plan.lv[level];
  const int k0 = slot0 - L.sel_off, k = k0 + grp;
  // (the slot always exists; a wave beyond the last slot reads the level's last one and leaves below)
  uint32_t e = (reinterpret_cast<const uint32_t*>(base + plan.sel_base) + L.sel_off)[min(k, L.sel_cap - 1)];
  static_assert(PS_ORB_MAX_LEVELS == 8, "the per-level counts are read as two int4");
  typedef const int32_t __attribute__((address_space(4))) * sel_scalar_ptr;     // wave-uniform address, written by the kernel before this one
  const sel_scalar_ptr scp = (sel_scalar_ptr)(uintptr_t)(base + plan.selcnt_off);
  int selc[8];
#pragma unroll
  for (int l = 0; l < 8; l++) selc[l] = scp[l];
  asm volatile("" :: "s"(selc[0]), "s"(selc[7]));   // (keeps the scalar loads on this side of the barrier: the compiler sinks them to their first use)
  reinterpret_cast<uint32_t*>(s_pat)[threadIdx.x] = t_pat;
  reinterpret_cast<uint32_t*>(s_icm)[threadIdx.x] = t_icm;
  __syncthreads();
  if (slot0 >= plan.sel_total) return;
  int offset = 0, total = 0, cnt = 0;
#pragma unroll
  for (int l = 0; l < PS_ORB_MAX_LEVELS; l++) {
    if (l < plan.nlevels) {
      const int c = selc[l];
      if (l < level) offset += c;
      if (l == level) cnt = c;
      total += c;
    }
  }
  if (slot0 == 0 && lane == 0) out_counts[img] = min(total, plan.kp_cap);
  if (k0 >= cnt) return;
  const int oi = offset + k;
  const bool valid = k < cnt && oi < plan.kp_cap;
  if (!valid) e = 19u | (19u << 12);       // an empty slot's row computes on a position whose loads stay inside the level; it stores nothing
  const int kx = e & 0xFFF, ky = (e >> 12) & 0xFFF, sc = e >> 24;

  // ---- the 39 x 39 neighbourhood of the blurred level that the steered pattern can reach (|coordinate| <= 18.4 before
  // rounding) goes to LDS with row-coalesced loads: the 512 byte gathers of the tests would otherwise touch ~30 cache lines
  // per load instruction.  Four lanes x 12 bytes per row, four rows per step, ten steps.  Issued first, consumed last.  The
  // rows land in LDS shifted to the patch's own first column (40-byte rows: 1560 bytes per keypoint, six workgroups per CU). ----
  __shared__ uint32_t patch_all[16][39 * 10];
  uint32_t* patch = patch_all[wv * 4 + grp];
  const int pshift = (kx - 19) & 3;
  const int r4 = l16 >> 2, c4 = l16 & 3;
  uint32_t tmp[10][3];
  {
    const uint32_t loff = L.blur_off + (uint32_t)(__mul24(ky - 19 + r4, L.bstride) + (kx - 19 - pshift) + 12 * c4);
    const uint32_t s4 = 4u * (uint32_t)L.bstride;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      // row 39 (i = 9, r4 = 3) does not exist in the neighbourhood: that lane repeats row 38 (never read)
      const uint32_t o = i < 9 ? loff + (uint32_t)i * s4 : loff + 9u * s4 - (r4 == 3 ? (uint32_t)L.bstride : 0u);
      const uint32_t* p = reinterpret_cast<const uint32_t*>(base + o);
      tmp[i][0] = p[0]; tmp[i][1] = p[1]; tmp[i][2] = p[2];
    }
  }
  // ---- IC_Angle: m10 = sum u*I, m01 = sum v*I over the disc.  32 rows x 4 groups of 8 columns = 128 items, 8 per lane: one
  // (unaligned) 8-byte load per item; disc mask and column weights (u + 15, as bytes) from a table, so an item costs four
  // v_dot4_u32_u8 and a multiply-add:  m10 = sum (u + 15) I - 15 sum I,   m01 = sum_rows v * (row sum) ----
  int m10, m01 = 0;
  {
    const uint32_t coff = L.plane_off + (uint32_t)(__mul24(PS_EDGE + ky - 15 + r4, L.stride) + PS_EDGE + kx - 15 + 8 * c4);
    const uint32_t s4 = 4u * (uint32_t)L.stride;
    uint2 pix[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pix[j] = *reinterpret_cast<const uint2*>(base + (coff + (uint32_t)j * s4));
    // the patch goes to LDS before the moments are computed: its 30 staging registers are free again by then
#pragma unroll
    for (int i = 0; i < 10; i++) {
      // the dword after the lane's three comes from the next lane of the quad
      const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tmp[i][0], 0xF9 /* quad_perm [1,2,3,3] */, 0xF, 0xF, false);
      const uint32_t o0 = __builtin_amdgcn_alignbyte(tmp[i][1], tmp[i][0], (uint32_t)pshift), o1 = __builtin_amdgcn_alignbyte(tmp[i][2], tmp[i][1], (uint32_t)pshift),
                     o2 = __builtin_amdgcn_alignbyte(nxt, tmp[i][2], (uint32_t)pshift);
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
      const uint32_t px = pix[j].x & mk.x, py = pix[j].y & mk.y;
      const uint32_t sr = __builtin_amdgcn_udot4(py, 0x01010101u, __builtin_amdgcn_udot4(px, 0x01010101u, 0u, false), false);
      acc = __builtin_amdgcn_udot4(py, whi, __builtin_amdgcn_udot4(px, wlo, acc, false), false);
      s0 += (int)sr;
      m01 += __mul24(v0 + 4 * j, (int)sr);            // (24-bit: a full 32-bit multiply issues at a quarter of the rate)
    }
    m10 = (int)acc - __mul24(15, s0);
  }
  m10 = row_sum_i32(m10);
  m01 = row_sum_i32(m01);
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // ---- steered BRIEF on the blurred level: lane l of the row handles tests 16 l .. 16 l + 15 ----
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float arad = __fmul_rn(angle, factorPI);
  double sn_d, cs_d;
  sincos_0_2pi((double)arad, sn_d, cs_d, c_sincos);
  const float a = (float)cs_d, b = (float)sn_d;
  wave_sync();
  // A point (x, y) of the pattern samples the blurred patch at row cvRound(x b + y a), column cvRound(x a - y b)
  // (ORBextractor.cc:115-120), every product and the sum rounded to float.  Two floats per instruction: {x b, x a} + {y a, -(y b)}
  // (negation is exact); cvRound by adding 1.5 * 2^23, whose float sum carries round-half-even(value) in its low mantissa
  // bits; the 24-bit multiply takes those bits as they are, the constants they drag along are subtracted from the base.
  const ds_f2 ba = {b, a}, anb = {a, -b};
  const unsigned long long ba64 = __builtin_bit_cast(unsigned long long, ba), anb64 = __builtin_bit_cast(unsigned long long, anb);
  const unsigned long long magic64 = 0x4B4000004B400000ull;  // {1.5 * 2^23, 1.5 * 2^23}
  typedef __attribute__((address_space(3))) const uint8_t lds_u8;
  // LDS byte address of patch pixel (0, 0) minus what the mantissa bits drag along
  const uint32_t kall = (uint32_t)(uintptr_t)(lds_u8*)reinterpret_cast<const uint8_t*>(patch) + (uint32_t)(19 * 40 + 19) -
                        (0x400000u * 40u + 0x4B400000u);
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
    // packed FP32 with the operand halves chosen by op_sel (the compiler scalarises this form): P = {x b, x a}, Q = {y a, -(y b)},
    // T = (P + Q) + magic - each instruction rounds on its own, as the separate float operations of the reference do.  The two
    // points of a test are interleaved so that no instruction reads the result of the one before it.
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
  if (!valid) return;
  reinterpret_cast<uint16_t*>(out_desc + ((size_t)img * plan.kp_cap + oi) * 32)[l16] = (uint16_t)bits;
  if (l16 == 0) {
    PsKeyPoint kp;
    kp.x = (float)kx;
    kp.y = (float)ky;
    if (level != 0) { kp.x = __fmul_rn(kp.x, L.scale); kp.y = __fmul_rn(kp.y, L.scale); }
    kp.size = L.kp_size;
    kp.angle = angle;
    kp.response = (float)(sc - 1);
    kp.octave = level;
    kp.class_id = -1;
    out_kps[(size_t)img * plan.kp_cap + oi] = kp;
  }
}

}  // namespace

// ---- launchers (called from orb_host.hip) --------------------------------------------------------
extern "C" void psk_orb_launch_pyramid(const OrbPlan* plan, int level, uint8_t* arena, const uint8_t* imgs,
                                       int img_stride, size_t img_pitch, const int4* tabs, int nimg,
                                       hipStream_t st) {
  const OrbLevel& L = plan->lv[level];
  const int groups = (L.w + 3 + 3) / 4;   // dword groups from padded column 16 to the last interior byte
  dim3 blk(64, 4), grd((groups + 63) / 64, (L.h + 4 * PYR_ROWS - 1) / (4 * PYR_ROWS), nimg);
  if (level == 0)
    hipLaunchKernelGGL(orb_pyramid_level<true>, grd, blk, 0, st, *plan, level, arena, imgs, img_stride, img_pitch, tabs);
  else
    hipLaunchKernelGGL(orb_pyramid_level<false>, grd, blk, 0, st, *plan, level, arena, imgs, img_stride, img_pitch, tabs);
}
// fused path: padded plane + blurred plane of one level in one launch
extern "C" void psk_orb_launch_level_fused(const OrbPlan* plan, int level, uint8_t* arena, const uint8_t* imgs,
                                           int img_stride, size_t img_pitch, const int4* tabs, int nimg, hipStream_t st) {
  const OrbLevel& L = plan->lv[level];
  const int PW = L.w + 2 * PS_EDGE, PH = L.h + 2 * PS_EDGE;
  const int tiles_x = (PW + LV_OWN_C - 1) / LV_OWN_C, tiles_y = (PH + LV_OWN_R - 1) / LV_OWN_R;
  const dim3 grd = PS_XCD_GRID(tiles_x * tiles_y, nimg);
  if (level == 0)
    hipLaunchKernelGGL(orb_level_fused<true>, grd, dim3(64 * LV_WAVES), 0, st, *plan, level, arena, imgs, img_stride, img_pitch, tabs, nimg, tiles_x);
  else
    hipLaunchKernelGGL(orb_level_fused<false>, grd, dim3(64 * LV_WAVES), 0, st, *plan, level, arena, imgs, img_stride, img_pitch, tabs, nimg, tiles_x);
}
extern "C" void psk_orb_launch_border(const OrbPlan* plan, uint8_t* arena, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(orb_border, dim3(plan->border_blocks, nimg), dim3(256), 0, st, *plan, arena);
}
extern "C" void psk_orb_launch_fast(const OrbPlan* plan, uint8_t* arena, int nimg, const uint8_t* masks, int mask_stride, size_t mask_pitch, const int4* tabs, hipStream_t st) {
  // LDS geometry from the largest cell window of the plan
  int mw = 0, mh = 0;
  for (int l = 0; l < plan->nlevels; l++) {
    mw = plan->lv[l].w_cell + 6 > mw ? plan->lv[l].w_cell + 6 : mw;
    mh = plan->lv[l].h_cell + 6 > mh ? plan->lv[l].h_cell + 6 : mh;
  }
  const int TR = mh;
  const int LCAP = (mw - 6) * (mh - 6);
  const int bpi = (plan->n_cells + 3) / 4;
  const uint32_t* celltab = reinterpret_cast<const uint32_t*>(tabs + plan->celltab_off);
  if (mw <= 38 && mh <= 48) {      // cells up to 32 px wide (30-px cells of the usual image sizes): 8 groups per row, 8 rows per step
    const FastLds F = fast_lds(40, TR, LCAP);
    hipLaunchKernelGGL((orb_fast_cells<40, 3, 48>), PS_XCD_GRID(bpi, nimg), dim3(FAST_T), (size_t)F.total * 4, st, *plan, arena, F,
                       nimg, bpi, masks, mask_stride, mask_pitch, celltab);
  } else {                         // up to PS_FAST_WIN: 16 groups per row, 4 rows per step
    const FastLds F = fast_lds(72, TR, LCAP);
    hipLaunchKernelGGL((orb_fast_cells<72, 4, 68>), PS_XCD_GRID(bpi, nimg), dim3(FAST_T), (size_t)F.total * 4, st, *plan, arena, F,
                       nimg, bpi, masks, mask_stride, mask_pitch, celltab);
  }
}
extern "C" void psk_orb_launch_quadtree(const OrbPlan* plan, uint8_t* arena, int nimg, hipStream_t st) {
  // node capacity a level needs: quota + 4 nodes, 4 * n_ini initial children, cells / 3 for the gather table
  int need[PS_ORB_MAX_LEVELS];
  for (int l = 0; l < plan->nlevels; l++) {
    const OrbLevel& L = plan->lv[l];
    need[l] = max(L.quota + 4, max(4 * L.n_ini, (L.n_cols * L.n_rows + 2) / 3));
  }
  // large batches: the top levels whose nodes fit 256 run with the small configuration (0.81 -> 0.73 ms per 1024 real-texture
  // images); a small batch does not fill the chip twice, there one launch of the 512-thread configuration is faster
  int split = plan->nlevels;
  if (nimg >= 256)
    while (split > 0 && need[split - 1] <= 256) split--;
  int big = 0;
  for (int l = 0; l < split; l++) big = max(big, need[l]);
  // grid: x = (level - first level) * nimg + image (level-major)
  if (split > 0) {
    const dim3 grid(split * nimg, 1);
    if (big <= 512) hipLaunchKernelGGL((orb_quadtree<512, QT_T, QT_KCAP512>), grid, dim3(QT_T), 0, st, *plan, arena, nimg, 0);
    else if (big <= 1024) hipLaunchKernelGGL((orb_quadtree<1024, QT_T, QT_KCAP>), grid, dim3(QT_T), 0, st, *plan, arena, nimg, 0);
    else hipLaunchKernelGGL((orb_quadtree<PS_QT_NCAP, QT_T, QT_KCAP>), grid, dim3(QT_T), 0, st, *plan, arena, nimg, 0);
  }
  if (split < plan->nlevels)
    hipLaunchKernelGGL((orb_quadtree<256, 256, 2048>), dim3((plan->nlevels - split) * nimg, 1), dim3(256), 0, st, *plan, arena, nimg, split);
}
extern "C" void psk_orb_launch_blur(const OrbPlan* plan, uint8_t* arena, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(orb_blur, dim3(plan->blur_blocks, nimg), dim3(256), 0, st, *plan, arena);
}
extern "C" int psk_orb_blur_rows() { return BLUR_ROWS; }
extern "C" void psk_orb_launch_describe(const OrbPlan* plan, uint8_t* arena, void* kps, uint8_t* desc,
                                        int32_t* counts, int nimg, hipStream_t st) {
  const int bpi = (plan->sel_total + 15) / 16;      // four waves x four keypoints per workgroup
  hipLaunchKernelGGL(orb_describe, PS_XCD_GRID(bpi, nimg), dim3(256), 0, st, *plan, arena,
                     (PsKeyPoint*)kps, desc, counts, nimg, bpi);
}
