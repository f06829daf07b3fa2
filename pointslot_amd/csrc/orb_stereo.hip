// Stereo matching of the two extractors' outputs on the device — Frame::ComputeStereoMatches
// (/root/reference/src/Frame.cc:2142-2316), SURVEY.md section 8f-1: it consumes both padded pyramids and both descriptor sets
// where they already are (HBM) and yields mvuRight / mvDepth.
//   st_match  : one wave per LEFT keypoint.  Candidate scan over the right keypoints (row-band test of the reference's
//               vRowIndices table evaluated on the fly, octave +-1, disparity range, Hamming; smallest (distance, index) by a
//               wave-min), then the 11 x 11 SAD over 11 shifts on the keypoint's pyramid level (integer-exact), parabola
//               sub-pixel fit and the disparity gates in the reference's float arithmetic
//   st_median : one workgroup per pair.  Rank selection of the median SAD, cut at 1.5f * 1.4f * median
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orb_plan.h"

namespace {

struct PsKeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
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
    rinfo[iR] = make_uint2(__float_as_uint(kr.x), (uint32_t)(min(max(minr + 1024, 0), 8191)) | ((uint32_t)(min(max(maxr + 1024, 0), 8191)) << 13) | ((uint32_t)kr.octave << 26));
    atomicAdd(&cnt[min(max((int)kr.y >> 3, 0), nb - 1)], 1);
  }
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int b = 0; b <= nb; b++) { const int c = cnt[b]; cnt[b] = acc; acc += c; } }
  __syncthreads();
  for (int b = tid; b <= nb; b += 256) boff[b] = cnt[b];
  for (int iR = tid; iR < Nr; iR += 256) {
    const int b = min(max((int)KR[iR].y >> 3, 0), nb - 1);
    bidx[cnt[b] + atomicAdd(&fill[b], 1)] = iR;
  }
}

__global__ __launch_bounds__(256) void st_match(OrbPlan plan, const StPair* pairs, float mb, float mbf) {
  __shared__ uint8_t rs_all[4][11 * 24];
  const StPair S = pairs[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int iL = blockIdx.x * 4 + wave;
  const int N = *S.cnt_l;
  if (iL >= N) return;
  uint8_t* rs = rs_all[wave];
  const PsKeyPoint* KL = reinterpret_cast<const PsKeyPoint*>(S.kps_l);
  const PsKeyPoint* KR = reinterpret_cast<const PsKeyPoint*>(S.kps_r);
  const PsKeyPoint kpL = KL[iL];
  float out_ur = -1.0f, out_depth = -1.0f;
  int out_sad = -1;
  const float minD = 0.f, maxD = __fdiv_rn(mbf, mb);
  const float uL = kpL.x, vL = kpL.y;
  const int levelL = kpL.octave;
  const int rowL = (int)vL;                      // vRowIndices[vL]
  const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
  uint32_t best = 0xFFFFFFFFu;
  if (!(maxU < 0)) {
    const uint4* dl = reinterpret_cast<const uint4*>(S.desc_l + (size_t)iL * 32);
    const uint4 a0 = dl[0], a1 = dl[1];
    // candidates: the buckets whose keypoints can have a band containing rowL (|y - rowL| <= 2 * scale[top] + 1)
    const int32_t* boff = reinterpret_cast<const int32_t*>(S.scratch);
    const int32_t* bidx = reinterpret_cast<const int32_t*>(S.scratch + PS_ST_OFF_BYTES);
    const uint2* rinfo = reinterpret_cast<const uint2*>(S.scratch + PS_ST_OFF_BYTES + PS_ST_CAP * 4);
    const int nb = min(PS_ST_MAXB, (plan.img_h >> 3) + 1);
    const float rmax = __fadd_rn(__fmul_rn(2.0f, plan.lv[plan.nlevels - 1].scale), 2.0f);
    const int b0 = min(max((int)floorf((float)rowL - rmax) >> 3, 0), nb - 1), b1 = min(max((int)ceilf((float)rowL + rmax) >> 3, 0), nb - 1);
    const int k_end = boff[b1 + 1];
    for (int k = boff[b0] + lane; k < k_end; k += 64) {
      const int iR = bidx[k];
      const uint2 ri = rinfo[iR];
      const int minr = (int)(ri.y & 0x1FFF) - 1024, maxr = (int)((ri.y >> 13) & 0x1FFF) - 1024, oct = (int)(ri.y >> 26);
      if (rowL < minr || rowL > maxr) continue;
      if (oct < levelL - 1 || oct > levelL + 1) continue;
      const float rx = __uint_as_float(ri.x);
      if (!(rx >= minU && rx <= maxU)) continue;
      const uint4* dr = reinterpret_cast<const uint4*>(S.desc_r + (size_t)iR * 32);
      const uint32_t d = (uint32_t)hamming256(a0, a1, dr[0], dr[1]);
      if (d < 100u) best = min(best, (d << 16) | (uint32_t)iR);   // bestDist starts at TH_HIGH, strict <, first wins
    }
  }
  best = wave_min_u32(best);
  const int bestDist = best == 0xFFFFFFFFu ? 100 : (int)(best >> 16);
  if (bestDist < 75) {   // thOrbDist = (TH_HIGH + TH_LOW) / 2
    const int bestIdxR = (int)(best & 0xFFFF);
    const OrbLevel L = plan.lv[levelL];
    const float uR0 = KR[bestIdxR].x;
    const float scaleduL = roundf(__fmul_rn(kpL.x, L.inv_scale));
    const float scaledvL = roundf(__fmul_rn(kpL.y, L.inv_scale));
    const float scaleduR0 = roundf(__fmul_rn(uR0, L.inv_scale));
    const float iniu = scaleduR0 + 5 - 5, endu = scaleduR0 + 5 + 5 + 1;
    if (!(iniu < 0 || endu >= (float)L.w)) {
      const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr = (int)scaleduR0;
      const uint8_t* pl = S.arena_l + L.plane_off + (size_t)(PS_EDGE + cy) * L.stride + PS_EDGE + cxl;
      const uint8_t* pr = S.arena_r + L.plane_off + (size_t)(PS_EDGE + cy) * L.stride + PS_EDGE + cxr;
      // right strip: rows -5..5, columns -10..10 around (cy, cxr)
      for (int q = lane; q < 11 * 21; q += 64) {
        const int i = q / 21, c = q - i * 21;
        rs[i * 24 + c] = pr[(ptrdiff_t)(i - 5) * L.stride + (c - 10)];
      }
      // this lane's (up to) two patch pixels
      const int p0 = lane, p1 = lane + 64;
      const int i0 = p0 / 11, j0 = p0 - i0 * 11, i1 = p1 / 11, j1 = p1 - i1 * 11;
      const int Lc = pl[0];
      const int l0 = (int)pl[(ptrdiff_t)(i0 - 5) * L.stride + (j0 - 5)] - Lc;
      const int l1 = p1 < 121 ? (int)pl[(ptrdiff_t)(i1 - 5) * L.stride + (j1 - 5)] - Lc : 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int dists[11];
#pragma unroll
      for (int s = 0; s < 11; s++) {   // incR = s - 5: right patch column j sits at strip column incR + j + 5
        const int Rc = rs[5 * 24 + s + 5];
        int acc = abs(l0 - ((int)rs[i0 * 24 + s + j0] - Rc));
        if (p1 < 121) acc += abs(l1 - ((int)rs[i1 * 24 + s + j1] - Rc));
        dists[s] = wave_sum_i32(acc);
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
          float bestuR = __fmul_rn(L.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
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
  if (lane == 0) { S.u_right[iL] = out_ur; S.depth[iL] = out_depth; S.sad[iL] = out_sad; }
}

__global__ __launch_bounds__(1024) void st_median(OrbPlan plan, const StPair* pairs) {
  __shared__ __attribute__((aligned(16))) uint32_t keys[PS_ST_CAP + 4];
  __shared__ int nkeys, median_sad;
  __shared__ int red[16];
  const StPair S = pairs[blockIdx.x];
  const int N = min(*S.cnt_l, PS_ST_CAP), tid = threadIdx.x;
  if (tid == 0) { nkeys = 0; median_sad = -1; }
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    const int sd = S.sad[i];
    if (sd >= 0) keys[atomicAdd(&nkeys, 1)] = ((uint32_t)sd << 12) | (uint32_t)i;   // order inside `keys` is irrelevant
  }
  __syncthreads();
  const int n = nkeys;
  if (n == 0) { if (tid == 0) *S.kept = 0; return; }
  if (tid < 4) keys[n + tid] = 0xFFFFFFFFu;      // padding of the vector reads: never smaller than a key
  __syncthreads();
  // the element of rank n/2 in (sad, index) order: sort(vDistIdx) then vDistIdx[size/2].first
  for (int e = tid; e < n; e += 1024) {
    const uint32_t k = keys[e];
    int rank = 0;
    for (int f = 0; f < n; f += 4) {
      const uint4 q = *reinterpret_cast<const uint4*>(&keys[f]);
      rank += (q.x < k ? 1 : 0) + (q.y < k ? 1 : 0) + (q.z < k ? 1 : 0) + (q.w < k ? 1 : 0);
    }
    if (rank == n / 2) median_sad = (int)(k >> 12);
  }
  __syncthreads();
  const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)median_sad);
  int kept = 0;
  for (int e = tid; e < n; e += 1024) {
    const uint32_t k = keys[e];
    const int i = (int)(k & 0xFFF);
    if ((float)(int)(k >> 12) < thDist) kept++;
    else { S.u_right[i] = -1.0f; S.depth[i] = -1.0f; }
  }
  kept = wave_sum_i32(kept);
  if ((tid & 63) == 0) red[tid >> 6] = kept;
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int w = 0; w < 16; w++) acc += red[w]; *S.kept = acc; }
}

}  // namespace

extern "C" void psk_stereo_launch(const OrbPlan* plan, const StPair* d_pairs, int npairs, int max_left, float mb, float mbf, hipStream_t st) {
  hipLaunchKernelGGL(st_bucket, dim3(npairs), dim3(256), 0, st, *plan, d_pairs);
  hipLaunchKernelGGL(st_match, dim3((max_left + 3) / 4, npairs), dim3(256), 0, st, *plan, d_pairs, mb, mbf);
  hipLaunchKernelGGL(st_median, dim3(npairs), dim3(1024), 0, st, *plan, d_pairs);
}
