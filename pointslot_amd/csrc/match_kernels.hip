// CDNA4 kernels for descriptor matching (256-bit ORB descriptors, Hamming distance).
//   bf_topk      : one wave per query — distances to every train descriptor of its problem (v_xor +
//                  v_bcnt), the 8 smallest (distance, index) pairs by repeated wave-min reduction
//   bf_resolve   : one wave per problem — the order-dependent greedy pass of SearchByBruceMatching
//                  (queries in order, trains already taken are skipped) over the top-8 lists, with an
//                  exact full-row rescan when a list is exhausted; rotation histogram + three maxima
//   hamming_matrix: full Nq x Nt distance matrix (bulk form of ORBmatcher::DescriptorDistance)
// Reference: /root/reference/src/ORBmatcher.cc:2043-2155, :2658-2699, :2704-2720.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "match_plan.h"

namespace {

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
  return v;
}

#define TOPK_T 256
// keys[w][j] = dist << 16 | j for the wave's current query (LDS), then 8 rounds of "smallest key greater
// than the previous one".
__global__ __launch_bounds__(TOPK_T) void bf_topk(const BfBlock* blocks, const BfProb* probs, const uint8_t* qdesc,
                                                   const uint8_t* tdesc, uint32_t* topk) {
  __shared__ uint32_t keys[TOPK_T / 64][PS_BF_MAX_TRAIN];
  const BfBlock blk = blocks[blockIdx.x];
  const BfProb P = probs[blk.prob];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint4* td = reinterpret_cast<const uint4*>(tdesc + (size_t)P.t_off * 32);
  for (int qi = blk.q_first + wave; qi < blk.q_first + blk.q_count; qi += TOPK_T / 64) {
    const uint4* qd = reinterpret_cast<const uint4*>(qdesc + (size_t)(P.q_off + qi) * 32);
    const uint4 a0 = qd[0], a1 = qd[1];
    for (int j = lane; j < P.nt; j += 64) {
      const int d = hamming256(a0, a1, td[2 * j], td[2 * j + 1]);
      keys[wave][j] = ((uint32_t)d << 16) | (uint32_t)j;
    }
    uint32_t prev = 0;
    bool first = true;
    uint32_t mine = 0xFFFFFFFFu;   // lane k keeps the k-th smallest
    for (int r = 0; r < PS_BF_TOPK; r++) {
      uint32_t m = 0xFFFFFFFFu;
      for (int j = lane; j < P.nt; j += 64) {
        const uint32_t k = keys[wave][j];
        if ((first || k > prev) && k < m) m = k;
      }
      m = wave_min_u32(m);
      if (lane == r) mine = m;
      if (m == 0xFFFFFFFFu) break;
      prev = m;
      first = false;
    }
    if (lane < PS_BF_TOPK) topk[(size_t)(P.q_off + qi) * PS_BF_TOPK + lane] = mine;
  }
}

__global__ __launch_bounds__(64) void bf_resolve(const BfProb* probs, const uint8_t* qdesc, const float* qang,
                                                  const uint8_t* qvalid, const uint8_t* tdesc, const float* tang,
                                                  const uint32_t* topk, int32_t* query_of_train, int32_t* nmatch,
                                                  float nn_ratio, int check_ori) {
  __shared__ uint32_t taken[PS_BF_MAX_TRAIN / 32];
  __shared__ uint8_t bin_of[PS_BF_MAX_TRAIN];
  __shared__ int hist[32];
  const BfProb P = probs[blockIdx.x];
  const int lane = threadIdx.x;
  int32_t* out = query_of_train + P.t_off;
  for (int j = lane; j < PS_BF_MAX_TRAIN / 32; j += 64) taken[j] = 0;
  if (lane < 32) hist[lane] = 0;
  for (int j = lane; j < P.nt; j += 64) out[j] = -1;
  __syncthreads();
  const uint4* td = reinterpret_cast<const uint4*>(tdesc + (size_t)P.t_off * 32);
  const float factor = 30 / 360.0f;   // HISTO_LENGTH / 360.0f
  int nm = 0;
  for (int q0 = 0; q0 < P.nq; q0 += 64) {
    const bool v = (q0 + lane < P.nq) && qvalid[P.q_off + q0 + lane] != 0;
    unsigned long long vm = __ballot(v);
    while (vm) {
      const int b = __ffsll((long long)vm) - 1;
      vm &= vm - 1;
      const int q = q0 + b;
      const uint32_t key = lane < PS_BF_TOPK ? topk[(size_t)(P.q_off + q) * PS_BF_TOPK + lane] : 0xFFFFFFFFu;
      const uint32_t k0 = (uint32_t)__shfl((int)key, 0);
      if ((k0 >> 16) > 50u) continue;   // best possible distance already above TH_LOW: can never match
      const uint32_t idx = key & 0xFFFF;
      const bool untaken = key != 0xFFFFFFFFu && !((taken[idx >> 5] >> (idx & 31)) & 1u);
      const unsigned long long um = __ballot(untaken);
      uint32_t best, second;
      if (__popcll(um) >= 2 || P.nt <= PS_BF_TOPK) {
        const int f = um ? __ffsll((long long)um) - 1 : -1;
        const unsigned long long um2 = um & (um - 1);
        const int s = um2 ? __ffsll((long long)um2) - 1 : -1;
        best = f >= 0 ? (uint32_t)__shfl((int)key, f) : (256u << 16);
        second = s >= 0 ? (uint32_t)__shfl((int)key, s) : (256u << 16);
      } else {
        // the list is exhausted: exact rescan of the row over untaken trains (two smallest keys)
        const uint4* qd = reinterpret_cast<const uint4*>(qdesc + (size_t)(P.q_off + q) * 32);
        const uint4 a0 = qd[0], a1 = qd[1];
        uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
        for (int j = lane; j < P.nt; j += 64) {
          if ((taken[j >> 5] >> (j & 31)) & 1u) continue;
          const uint32_t k = ((uint32_t)hamming256(a0, a1, td[2 * j], td[2 * j + 1]) << 16) | (uint32_t)j;
          if (k < m1) { m2 = m1; m1 = k; } else if (k < m2) m2 = k;
        }
        best = wave_min_u32(m1);
        const uint32_t cand = (m1 == best) ? m2 : m1;
        second = wave_min_u32(cand);
        if (best == 0xFFFFFFFFu) best = 256u << 16;
        if (second == 0xFFFFFFFFu) second = 256u << 16;
      }
      const int d1 = (int)(best >> 16), d2 = (int)(second >> 16);
      if (d1 <= 50 && (float)d1 < __fmul_rn(nn_ratio, (float)d2)) {
        const int bi = (int)(best & 0xFFFF);
        if (lane == 0) {
          taken[bi >> 5] |= 1u << (bi & 31);
          out[bi] = q;
          if (check_ori) {
            float rot = __fsub_rn(qang[P.q_off + q], tang[P.t_off + bi]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == 30) bin = 0;
            hist[bin]++;
            bin_of[bi] = (uint8_t)bin;
          }
        }
        nm++;
        __syncthreads();   // single wave: orders the LDS updates before the next query's reads
      }
    }
  }
  __syncthreads();
  if (check_ori) {
    // ComputeThreeMaxima (ORBmatcher.cc:2658-2699), evaluated redundantly by every lane
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < 30; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
      else if (s > max3) { max3 = s; i3 = i; }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
    else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
    int removed = 0;
    for (int j = lane; j < P.nt; j += 64) {
      if (out[j] >= 0) {
        const int b = bin_of[j];
        if (b != i1 && b != i2 && b != i3) { out[j] = -1; removed++; }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) removed += __shfl_xor(removed, d);
    nm -= removed;
  }
  if (lane == 0) nmatch[blockIdx.x] = nm;
}

__global__ __launch_bounds__(256) void hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= nt) return;
  const uint4* qd = reinterpret_cast<const uint4*>(q + (size_t)i * 32);
  const uint4* td = reinterpret_cast<const uint4*>(t + (size_t)j * 32);
  out[(size_t)i * nt + j] = (uint16_t)hamming256(qd[0], qd[1], td[0], td[1]);
}

}  // namespace

extern "C" void psk_bf_launch(const BfBlock* blocks, int nblocks, const BfProb* probs, int nprob, const uint8_t* qdesc,
                              const float* qang, const uint8_t* qvalid, const uint8_t* tdesc, const float* tang,
                              uint32_t* topk, int32_t* out, int32_t* nmatch, float nn_ratio, int check_ori,
                              hipStream_t st) {
  if (nblocks > 0) hipLaunchKernelGGL(bf_topk, dim3(nblocks), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk);
  hipLaunchKernelGGL(bf_resolve, dim3(nprob), dim3(64), 0, st, probs, qdesc, qang, qvalid, tdesc, tang, topk, out,
                     nmatch, nn_ratio, check_ori);
}
extern "C" void psk_hamming_matrix_launch(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out, hipStream_t st) {
  hipLaunchKernelGGL(hamming_matrix, dim3((nt + 255) / 256, nq), dim3(256), 0, st, q, nq, t, nt, out);
}
