// Stereo matching of the two extractors' outputs on the device — Frame::ComputeStereoMatches
// (/root/reference/src/Frame.cc:2142-2316), SURVEY.md section 8f-1: it consumes both padded pyramids and both descriptor sets
// where they already are (HBM) and yields mvuRight / mvDepth.
//   st_match  : FOUR left keypoints per wave, one per 16-lane DPP row.  Candidate scan over the right keypoints (row-band test of
//               the reference's vRowIndices table evaluated on the fly, octave +-1, disparity range, Hamming; smallest (distance,
//               index) by a row-min), then the 11 x 11 SAD over 11 shifts on the keypoint's pyramid level (integer-exact: a lane
//               per patch row, both rows in registers, v_sad_u16 on packed pairs), parabola sub-pixel fit and the disparity
//               gates in the reference's float arithmetic
//   st_median : one workgroup per pair.  Rank selection of the median SAD, cut at 1.5f * 1.4f * median
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orb_plan.h"

namespace {

// The pointers of a StPair come out of memory, where the compiler cannot see their address space and would use FLAT loads
// (both memory pipelines, half the issue rate); they all point to device memory.
#define ST_G(T) const __attribute__((address_space(1))) T
#define ST_GM(T) __attribute__((address_space(1))) T
template <typename T> __device__ __forceinline__ ST_G(T)* st_g(const T* p) { return (ST_G(T)*)p; }
template <typename T> __device__ __forceinline__ ST_GM(T)* st_gm(T* p) { return (ST_GM(T)*)p; }

typedef uint32_t st_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u2 __attribute__((ext_vector_type(2)));
struct PsKeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

__device__ __forceinline__ int hamming256(const st_u4 a0, const st_u4 a1, const st_u4 b0, const st_u4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}
// wave-wide reductions on DPP lanes (xor-1, xor-2, half-row mirror, row mirror) + four readlanes; all 64 lanes active
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// Row table of the right keypoints (the reference's vRowIndices, Frame.cc:2153-2169) in bucket form: the keypoints are grouped by
// 8-row buckets of their y; a left keypoint on row v only has to look at the buckets that can hold a band [minr, maxr]
// containing v.  One workgroup per pair.
__global__ __launch_bounds__(256) void st_bucket(OrbPlan plan, const StPair* pairs) {
  __shared__ int cnt[PS_ST_MAXB + 1], fill[PS_ST_MAXB];
  const StPair S = pairs[blockIdx.x];
  const int Nr = min(*S.cnt_r, PS_ST_CAP), tid = threadIdx.x;
  const PsKeyPoint* KR = reinterpret_cast<const PsKeyPoint*>(S.kps_r);
  int32_t* boff = reinterpret_cast<int32_t*>(S.scratch);
  int32_t* bidx = reinterpret_cast<int32_t*>(S.scratch + PS_ST_OFF_BYTES);
  uint2* rinfo = reinterpret_cast<uint2*>(S.scratch + PS_ST_OFF_BYTES + PS_ST_CAP * 4);
  const int nb = min(PS_ST_MAXB, (plan.img_h >> 3) + 1);
  for (int b = tid; b <= PS_ST_MAXB; b += 256) { cnt[b] = 0; if (b < PS_ST_MAXB) fill[b] = 0; }
  __syncthreads();
  for (int iR = tid; iR < Nr; iR += 256) {
    const PsKeyPoint kr = KR[iR];
    const float r = __fmul_rn(2.0f, plan.lv[kr.octave].scale);
    const int maxr = (int)ceilf(__fadd_rn(kr.y, r)), minr = (int)floorf(__fsub_rn(kr.y, r));
    // bands are stored with an offset of 1024 so that rows a little outside the image stay non-negative
    (void)maxr; (void)minr;
    atomicAdd(&cnt[min(max((int)kr.y >> 3, 0), nb - 1)], 1);
  }
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int b = 0; b <= nb; b++) { const int c = cnt[b]; cnt[b] = acc; acc += c; } }
  __syncthreads();
  for (int b = tid; b <= nb; b += 256) boff[b] = cnt[b];
  for (int iR = tid; iR < Nr; iR += 256) {
    const PsKeyPoint kr = KR[iR];
    const int b = min(max((int)kr.y >> 3, 0), nb - 1);
    const int pos = cnt[b] + atomicAdd(&fill[b], 1);
    const float r = __fmul_rn(2.0f, plan.lv[kr.octave].scale);
    const int maxr = (int)ceilf(__fadd_rn(kr.y, r)), minr = (int)floorf(__fsub_rn(kr.y, r));
    bidx[pos] = iR;
    // the band record sits at the keypoint's position in bucket order (st_match reads both arrays by position); bands are
    // stored with an offset of 1024 so that rows a little outside the image stay non-negative
    rinfo[pos] = make_uint2(__float_as_uint(kr.x), (uint32_t)(min(max(minr + 1024, 0), 8191)) | ((uint32_t)(min(max(maxr + 1024, 0), 8191)) << 13) | ((uint32_t)kr.octave << 26));
  }
}

typedef unsigned short st_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {     // min over the 16 lanes of a DPP row, left in every lane of the row
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
  return v;
}
__device__ __forceinline__ int row_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
  return v;
}
// bytes (b, b + 1) of the 8-byte pair {hi, lo} as packed u16; a byte index of 0x0c yields zero
#define ST_PAIR(hi, lo, b0, b1) __builtin_amdgcn_perm(hi, lo, (uint32_t)(b0) | 0x0c000c00u | ((uint32_t)(b1) << 16))

__global__ __launch_bounds__(256) void st_match(OrbPlan plan, const StPair* pairs, float mb, float mbf, int npairs) {
  // the level fields a keypoint needs, indexed by its octave (a lane-dependent index into the kernel arguments would go through memory anyway)
  __shared__ float s_scale[PS_ORB_MAX_LEVELS], s_inv[PS_ORB_MAX_LEVELS];
  __shared__ int s_w[PS_ORB_MAX_LEVELS], s_stride[PS_ORB_MAX_LEVELS];
  __shared__ uint32_t s_plane[PS_ORB_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < PS_ORB_MAX_LEVELS; l++)
    if ((int)threadIdx.x == l) { s_scale[l] = plan.lv[l].scale; s_inv[l] = plan.lv[l].inv_scale; s_w[l] = plan.lv[l].w; s_stride[l] = plan.lv[l].stride; s_plane[l] = plan.lv[l].plane_off; }
  __syncthreads();
  // one pair per XCD at a time (workgroup b runs on XCD b % 8; the grid is 8 * blocks wide): the candidate lists, the right image's
  // descriptors and the planes of a pair are then fetched into one L2 instead of eight
  const int pair = (int)blockIdx.y * 8 + ((int)blockIdx.x & 7);
  if (pair >= npairs) return;
  const StPair S = pairs[pair];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
  const int N = *st_g(S.cnt_l);
  const int iL0 = ((int)blockIdx.x >> 3) * 16 + wave * 4;
  if (iL0 >= N) return;
  const int iL = iL0 + grp;
  const bool live = iL < N;
  ST_G(PsKeyPoint)* KL = st_g(reinterpret_cast<const PsKeyPoint*>(S.kps_l));
  const int iLc = live ? iL : N - 1;
  PsKeyPoint kpL;
  kpL.x = KL[iLc].x; kpL.y = KL[iLc].y; kpL.octave = KL[iLc].octave;
  float out_ur = -1.0f, out_depth = -1.0f;
  int out_sad = -1;
  const float minD = 0.f, maxD = __fdiv_rn(mbf, mb);
  const float uL = kpL.x, vL = kpL.y;
  const int levelL = kpL.octave;
  const int rowL = (int)vL;                      // vRowIndices[vL]
  const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
  uint32_t best = 0xFFFFFFFFu;
  float best_rx = 0.f;
  if (live && !(maxU < 0)) {
    ST_G(st_u4)* dl = st_g(reinterpret_cast<const st_u4*>(S.desc_l + (size_t)iL * 32));
    const st_u4 a0 = dl[0], a1 = dl[1];
    // candidates: the buckets whose keypoints can have a band containing rowL (|y - rowL| <= 2 * scale[top] + 1)
    ST_G(int32_t)* boff = st_g(reinterpret_cast<const int32_t*>(S.scratch));
    ST_G(int32_t)* bidx = st_g(reinterpret_cast<const int32_t*>(S.scratch + PS_ST_OFF_BYTES));
    ST_G(st_u2)* rinfo = st_g(reinterpret_cast<const st_u2*>(S.scratch + PS_ST_OFF_BYTES + PS_ST_CAP * 4));
    const int nb = min(PS_ST_MAXB, (plan.img_h >> 3) + 1);
    const float rmax = __fadd_rn(__fmul_rn(2.0f, plan.lv[plan.nlevels - 1].scale), 2.0f);
    const int b0 = min(max((int)floorf((float)rowL - rmax) >> 3, 0), nb - 1), b1 = min(max((int)ceilf((float)rowL + rmax) >> 3, 0), nb - 1);
    const int k_end = boff[b1 + 1];
    const uint32_t rowk = (uint32_t)(rowL + 1024);
    // the band records are walked four per lane and step, all four requested before the first is looked at (a record is a round
    // trip to L2; few pass the row / octave / column tests and go on to a descriptor)
    for (int k = boff[b0] + l16; k < k_end; k += 64) {
      st_u2 ri[4]; int iR[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int ku = min(k + 16 * u, k_end - 1); ri[u] = rinfo[ku]; iR[u] = bidx[ku]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k + 16 * u >= k_end) continue;
        const uint32_t minr = ri[u].y & 0x1FFF, maxr = (ri[u].y >> 13) & 0x1FFF;
        const int oct = (int)(ri[u].y >> 26);
        if (rowk < minr || rowk > maxr) continue;
        if ((uint32_t)(oct - levelL + 1) > 2u) continue;
        const float rx = __uint_as_float(ri[u].x);
        if (!(rx >= minU && rx <= maxU)) continue;
        ST_G(st_u4)* dr = st_g(reinterpret_cast<const st_u4*>(S.desc_r + (size_t)iR[u] * 32));
        const uint32_t d = (uint32_t)hamming256(a0, a1, dr[0], dr[1]);
        if (d < 100u) {                                  // bestDist starts at TH_HIGH, strict <, first wins
          const uint32_t key = (d << 16) | (uint32_t)iR[u];
          if (key < best) { best = key; best_rx = rx; }  // the record carries the right keypoint's x: no second look-up for the winner
        }
      }
    }
  }
  const uint32_t lane_best = best;
  best = row_min_u32(best);
  // the x of the best candidate, out of the lane that found it (keys are unique: they end in the candidate's index)
  float uR0 = 0.f;
  {
    const uint32_t owners = (uint32_t)(__ballot(lane_best == best && best != 0xFFFFFFFFu) >> (lane & 48)) & 0xFFFFu;
    uR0 = __shfl(best_rx, (lane & 48) + (owners ? __ffs((int)owners) - 1 : 0));
  }
  const int bestDist = best == 0xFFFFFFFFu ? 100 : (int)(best >> 16);
  if (bestDist < 75) {   // thOrbDist = (TH_HIGH + TH_LOW) / 2
    const float Lscale = s_scale[levelL], Linv = s_inv[levelL];
    const int Lw = s_w[levelL], Lstride = s_stride[levelL];
    const uint32_t Lplane = s_plane[levelL];
    const float scaleduL = roundf(__fmul_rn(kpL.x, Linv));
    const float scaledvL = roundf(__fmul_rn(kpL.y, Linv));
    const float scaleduR0 = roundf(__fmul_rn(uR0, Linv));
    const float iniu = scaleduR0 + 5 - 5, endu = scaleduR0 + 5 + 5 + 1;
    if (!(iniu < 0 || endu >= (float)Lw)) {
      const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr = (int)scaleduR0;
      // lane i < 11 of the row owns patch row i - 5: its 11 left pixels (columns -5 .. 5) and the 21 right pixels (columns
      // -10 .. 10) that the 11 shifts slide over, in registers; every lane also reads the centre row for the centre pixels.
      const int i = min(l16, 10);
      ST_G(uint8_t)* al = st_g(S.arena_l);
      ST_G(uint8_t)* ar = st_g(S.arena_r);
      ST_G(uint32_t)* pl = (ST_G(uint32_t)*)(al + (Lplane + (uint32_t)((PS_EDGE + cy + i - 5) * Lstride + PS_EDGE + cxl - 5)));
      ST_G(uint32_t)* pr = (ST_G(uint32_t)*)(ar + (Lplane + (uint32_t)((PS_EDGE + cy + i - 5) * Lstride + PS_EDGE + cxr - 10)));
      ST_G(uint32_t)* prc = (ST_G(uint32_t)*)(ar + (Lplane + (uint32_t)((PS_EDGE + cy) * Lstride + PS_EDGE + cxr - 10)));
      uint32_t lw[3], rw[6], cw[6];
#pragma unroll
      for (int q = 0; q < 3; q++) lw[q] = pl[q];
#pragma unroll
      for (int q = 0; q < 6; q++) { rw[q] = pr[q]; cw[q] = prc[q]; }
      const uint32_t Lc = al[Lplane + (uint32_t)((PS_EDGE + cy) * Lstride + PS_EDGE + cxl)];
      // |(L - Lc) - (R - Rc)| on packed u16 with both sides offset by 256: a = L + (256 - Lc), b = R + (256 - Rc)
      const uint32_t biasA = (256u - Lc) * 0x00010001u;
      uint32_t a[6];
#pragma unroll
      for (int q = 0; q < 5; q++) a[q] = ST_PAIR(lw[q >> 1], lw[q >> 1], (2 * q) & 3, (2 * q + 1) & 3) + biasA;   // an even pair shares a dword
      a[5] = ST_PAIR(0u, lw[2], 2, 0x0c) + (biasA & 0xFFFFu);                        // pixel 10 alone: the other half stays zero on both sides
      // right row as packed pairs at both alignments: E[k] = (R[2k], R[2k+1]), O[k] = (R[2k+1], R[2k+2])
      uint32_t E[11], O[10];
#pragma unroll
      for (int q = 0; q < 11; q++) {
        const int b0i = 2 * q, b1i = 2 * q + 1;
        E[q] = ST_PAIR(rw[b0i >> 2], rw[b0i >> 2], b0i & 3, q == 10 ? 0x0c : (b1i & 3));     // both bytes of an even pair share a dword
      }
#pragma unroll
      for (int q = 0; q < 10; q++) {
        const int b0i = 2 * q + 1, b1i = 2 * q + 2;
        O[q] = ST_PAIR(rw[b1i >> 2], rw[b0i >> 2], b0i & 3, (b1i & 3) + ((b1i >> 2) != (b0i >> 2) ? 4 : 0));
      }
      int dists[11];
#pragma unroll
      for (int s = 0; s < 11; s++) {   // incR = s - 5: right patch column j sits at strip column s + j
        const uint32_t Rc = (cw[(s + 5) >> 2] >> (8 * ((s + 5) & 3))) & 0xFFu;
        const uint32_t biasB = (256u - Rc) * 0x00010001u;
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
          const uint32_t r2 = (s & 1) ? O[(s - 1) / 2 + q] : E[s / 2 + q];
          acc = __builtin_amdgcn_sad_u16(a[q], r2 + biasB, acc);
        }
        const uint32_t rl = ((s & 1) ? O[(s - 1) / 2 + 5] : E[s / 2 + 5]) & 0xFFFFu;   // R[s + 10] alone
        acc = __builtin_amdgcn_sad_u16(a[5], rl + (biasB & 0xFFFFu), acc);
        dists[s] = row_sum_i32(l16 < 11 ? (int)acc : 0);
      }
      int bestS = 0x7fffffff, bestinc = 0;
#pragma unroll
      for (int s = 0; s < 11; s++)
        if (dists[s] < bestS) { bestS = dists[s]; bestinc = s - 5; }
      if (bestinc != -5 && bestinc != 5) {
        float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int s = 1; s < 10; s++)
          if (s - 5 == bestinc) { d1 = (float)dists[s - 1]; d2 = (float)dists[s]; d3 = (float)dists[s + 1]; }
        const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR = __fmul_rn(Lscale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
          float disparity = __fsub_rn(uL, bestuR);
          if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
              disparity = 0.01f;
              bestuR = (float)((double)uL - 0.01);
            }
            out_depth = __fdiv_rn(mbf, disparity);
            out_ur = bestuR;
            out_sad = bestS;
          }
        }
      }
    }
  }
  if (l16 == 0 && live) { st_gm(S.u_right)[iL] = out_ur; st_gm(S.depth)[iL] = out_depth; st_gm(S.sad)[iL] = out_sad; }
}

#ifndef ST_MED_T
#define ST_MED_T 256      // (1024 threads - sixteen wave slots of ONE CU free at the same moment - waited 200 us for its turn beside the other
                          // lockstep groups' kernels; alone it ran 29 us)
#endif
__global__ __launch_bounds__(ST_MED_T) void st_median(OrbPlan plan, const StPair* pairs) {
  __shared__ __attribute__((aligned(16))) uint32_t keys[PS_ST_CAP + 4];
  __shared__ int nkeys, median_sad;
  __shared__ int red[ST_MED_T / 64];
  const StPair S = pairs[blockIdx.x];
  const int N = min(*S.cnt_l, PS_ST_CAP), tid = threadIdx.x;
  if (tid == 0) { nkeys = 0; median_sad = -1; }
  __syncthreads();
  for (int i = tid; i < N; i += ST_MED_T) {
    const int sd = S.sad[i];
    if (sd >= 0) keys[atomicAdd(&nkeys, 1)] = ((uint32_t)sd << 12) | (uint32_t)i;   // order inside `keys` is irrelevant
  }
  __syncthreads();
  const int n = nkeys;
  if (n == 0) { if (tid == 0) *S.kept = 0; return; }
  if (tid < 4) keys[n + tid] = 0xFFFFFFFFu;      // padding of the vector reads: never smaller than a key
  __syncthreads();
  // the SAD of the element of rank n/2 in (sad, index) order - sort(vDistIdx) then vDistIdx[size/2].first - is the value of rank
  // n/2 of the SAD multiset (the index only orders equal values): a two-level radix selection on the 16-bit value (an 11 x 11
  // patch sums to at most 61 710) instead of counting ranks, O(n) instead of O(n^2)
  {
    __shared__ int hist[256];
    __shared__ int sel_bin, sel_rank;
    int want = n / 2;
    for (int level = 0; level < 2; level++) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const int hi_bin = level == 1 ? sel_bin : 0;
      for (int e = tid; e < n; e += ST_MED_T) {
        const uint32_t sd = keys[e] >> 12;
        if (level == 0) atomicAdd(&hist[sd >> 8], 1);
        else if ((int)(sd >> 8) == hi_bin) atomicAdd(&hist[sd & 255u], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 256; b++) { if (acc + hist[b] > want) break; acc += hist[b]; }
        if (level == 0) { sel_bin = b; sel_rank = want - acc; }
        else median_sad = (hi_bin << 8) | b;
      }
      __syncthreads();
      want = sel_rank;
    }
  }
  const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)median_sad);
  int kept = 0;
  for (int e = tid; e < n; e += ST_MED_T) {
    const uint32_t k = keys[e];
    const int i = (int)(k & 0xFFF);
    if ((float)(int)(k >> 12) < thDist) kept++;
    else { S.u_right[i] = -1.0f; S.depth[i] = -1.0f; }
  }
  kept = wave_sum_i32(kept);
  if ((tid & 63) == 0) red[tid >> 6] = kept;
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int w = 0; w < ST_MED_T / 64; w++) acc += red[w]; *S.kept = acc; }
}

}  // namespace

extern "C" void psk_stereo_launch(const OrbPlan* plan, const StPair* d_pairs, int npairs, int max_left, float mb, float mbf, hipStream_t st) {
  hipLaunchKernelGGL(st_bucket, dim3(npairs), dim3(256), 0, st, *plan, d_pairs);
  hipLaunchKernelGGL(st_match, dim3(((max_left + 15) / 16) * 8, (npairs + 7) / 8), dim3(256), 0, st, *plan, d_pairs, mb, mbf, npairs);
  hipLaunchKernelGGL(st_median, dim3(npairs), dim3(ST_MED_T), 0, st, *plan, d_pairs);
}
