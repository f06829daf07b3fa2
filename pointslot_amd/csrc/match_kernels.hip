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

// wave-wide unsigned minimum on DPP lanes (xor-1, xor-2, half-row mirror, row mirror) + four readlanes: no LDS round trips.
// All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

#define TOPK_T 256
#define BF_SMALL_NT PS_BF_SMALL_NT
// unsigned minimum inside a 16-lane DPP row (every lane of the row gets the result)
__device__ __forceinline__ uint32_t bf_row_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
  return v;
}
// Problems with at most 256 train descriptors (an object's features in the tracker: ~150): FOUR queries per wave, one per 16-lane
// row, the row's keys in registers (16 per lane) - no LDS, so the occupancy is not the 64 KB key store's two workgroups per CU.
// Same keys, same order of the eight results as bf_topk.
// Two instances: NT = 256 (16 key registers) for the usual object, NT = 512 for the close ones.
template <int NT, int LO>
__global__ __launch_bounds__(TOPK_T) void bf_topk_small(const BfBlock* blocks, const BfProb* probs, const uint8_t* qdesc,
                                                         const uint8_t* tdesc, uint32_t* topk, const int32_t* count) {
  const int nb = count ? *count : (int)gridDim.x;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  for (int bi = blockIdx.x; bi < nb; bi += gridDim.x) {
    const BfBlock blk = blocks[bi];
    const BfProb P = probs[blk.prob];
    if (P.nt > NT || P.nt <= LO) continue;
    const uint4* td = reinterpret_cast<const uint4*>(tdesc + (size_t)P.t_off * 32);
    for (int q0 = 0; q0 < blk.q_count; q0 += TOPK_T / 16) {
      const bool live = q0 + grp < blk.q_count;
      const int qi = blk.q_first + min(q0 + grp, blk.q_count - 1);
      const uint4* qd = reinterpret_cast<const uint4*>(qdesc + (size_t)(P.q_off + qi) * 32);
      const uint4 a0 = qd[0], a1 = qd[1];
      uint32_t key[NT / 16];
      const int nu = (P.nt + 15) >> 4;            // key registers in use (uniform): the loops below leave at nu
#pragma unroll
      for (int u = 0; u < NT / 16; u++) {
        const int j = l16 + 16 * u;
        key[u] = 0xFFFFFFFFu;
        if (u < nu && j < P.nt) key[u] = ((uint32_t)hamming256(a0, a1, td[2 * j], td[2 * j + 1]) << 16) | (uint32_t)j;
      }
      uint32_t prev = 0, mine = 0xFFFFFFFFu;
      bool first = true;
#pragma unroll 1
      for (int r = 0; r < PS_BF_TOPK; r++) {
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < NT / 16; u++) {
          if (u >= nu) break;
          m = ((first || key[u] > prev) && key[u] < m) ? key[u] : m;
        }
        m = bf_row_min_u32(m);
        if (l16 == r) mine = m;
        prev = m;                    // an exhausted row stays at 0xFFFFFFFF: nothing is greater
        first = false;
      }
      if (live && l16 < PS_BF_TOPK) topk[(size_t)(P.q_off + qi) * PS_BF_TOPK + l16] = mine;
    }
  }
}
// keys[w][j] = dist << 16 | j for the wave's current query (LDS), then 8 rounds of "smallest key greater
// than the previous one".
// `count` (nullable): the number of entries of `blocks` when the table was built on the device (the lockstep tracker); the grid then
// loops over it.  Host-built tables are launched with one workgroup per entry.
__global__ __launch_bounds__(TOPK_T) void bf_topk(const BfBlock* blocks, const BfProb* probs, const uint8_t* qdesc,
                                                   const uint8_t* tdesc, uint32_t* topk, const int32_t* count) {
  __shared__ uint32_t keys[TOPK_T / 64][PS_BF_MAX_TRAIN];
  const int nb = count ? *count : (int)gridDim.x;
  if (count && count[1] == 0) return;             // device-built table: count[1] = some problem has more than BF_SMALL_NT trains
  for (int bi = blockIdx.x; bi < nb; bi += gridDim.x) {
  const BfBlock blk = blocks[bi];
  const BfProb P = probs[blk.prob];
  if (P.nt <= BF_SMALL_NT) continue;              // bf_topk_small's problem
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint4* td = reinterpret_cast<const uint4*>(tdesc + (size_t)P.t_off * 32);
  for (int qi = blk.q_first + wave; qi < blk.q_first + blk.q_count; qi += TOPK_T / 64) {
    const uint4* qd = reinterpret_cast<const uint4*>(qdesc + (size_t)(P.q_off + qi) * 32);
    const uint4 a0 = qd[0], a1 = qd[1];
    for (int j0 = lane; j0 < P.nt; j0 += 256) {       // four train descriptors per lane and step, requested together
      uint4 b0[4], b1[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int j = min(j0 + 64 * u, P.nt - 1); b0[u] = td[2 * j]; b1[u] = td[2 * j + 1]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int j = j0 + 64 * u;
        if (j < P.nt) keys[wave][j] = ((uint32_t)hamming256(a0, a1, b0[u], b1[u]) << 16) | (uint32_t)j;
      }
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
}

#define BF_RES_TANG 1024      // train angles kept in LDS by bf_resolve (larger problems read them from memory)
#define BF_CLAIM 512
__global__ __launch_bounds__(64) void bf_resolve(const BfProb* probs, const uint8_t* qdesc, const float* qang,
                                                  const uint8_t* qvalid, const uint8_t* tdesc, const float* tang,
                                                  const uint32_t* topk, int32_t* query_of_train, int32_t* nmatch,
                                                  float nn_ratio, int check_ori, int nprob_total) {
  __shared__ uint32_t taken[PS_BF_MAX_TRAIN / 32];
  __shared__ uint8_t bin_of[PS_BF_MAX_TRAIN];
  __shared__ int hist[32];
  __shared__ float tang_s[BF_RES_TANG];
  __shared__ uint32_t claim[BF_CLAIM];   // hashed train -> earliest pending lane of the chunk whose best it is
  // (r05) problem <-> workgroup rotated inside every group of eight: workgroup b runs on XCD b % 8, and the tracker's problems are
  // (sequence, detection slot) with eight slots of which the first two are live - unrotated, they all ran on XCDs 0 and 1
  const int pbi = ((int)blockIdx.x & ~7) | (((int)blockIdx.x + ((int)blockIdx.x >> 3)) & 7);
  if (pbi >= nprob_total) return;
  const BfProb P = probs[pbi];
  const int lane = threadIdx.x;
  if (P.nt == 0 || P.nq == 0) {   // nothing to match (bf_topk was not run for this problem)
    for (int j = lane; j < P.nt; j += 64) query_of_train[P.t_off + j] = -1;
    if (lane == 0) nmatch[pbi] = 0;
    return;
  }
  int32_t* out = query_of_train + P.t_off;
  for (int j = lane; j < PS_BF_MAX_TRAIN / 32; j += 64) taken[j] = 0;
  if (lane < 32) hist[lane] = 0;
  for (int j = lane; j < BF_CLAIM; j += 64) claim[j] = 0xFFFFFFFFu;
  for (int j = lane; j < P.nt; j += 64) out[j] = -1;
  __syncthreads();
  const uint4* td = reinterpret_cast<const uint4*>(tdesc + (size_t)P.t_off * 32);
  const float factor = 30 / 360.0f;   // HISTO_LENGTH / 360.0f
  // The queries are resolved one after the other (the reference's order decides who gets a train).  What a query needs from memory
  // - its eight keys, its angle, the matched train's angle - is fetched per chunk of 64 queries (lane = query) and per problem (train
  // angles in LDS), not per query: the serial loop then runs on registers and LDS instead of one L2 round trip or two per query.
  const bool tang_lds = check_ori && P.nt <= BF_RES_TANG;
  if (tang_lds)
    for (int j = lane; j < P.nt; j += 64) tang_s[j] = tang[P.t_off + j];
  __syncthreads();
  int nm = 0;
  for (int q0 = 0; q0 < P.nq; q0 += 64) {
    const bool v = (q0 + lane < P.nq) && qvalid[P.q_off + q0 + lane] != 0;
    uint32_t kk[PS_BF_TOPK];
    float qa = 0.f;
    {
      const int qc = min(q0 + lane, P.nq - 1);
      const uint4* kp = reinterpret_cast<const uint4*>(topk + (size_t)(P.q_off + qc) * PS_BF_TOPK);
      const uint4 ka = kp[0], kb = kp[1];
      kk[0] = ka.x; kk[1] = ka.y; kk[2] = ka.z; kk[3] = ka.w; kk[4] = kb.x; kk[5] = kb.y; kk[6] = kb.z; kk[7] = kb.w;
      if (check_ori) qa = qang[P.q_off + qc];
    }
    // ---- the 64 queries of the chunk, lane = query.  A query's outcome depends on the earlier ones only through the trains they take,
    // and it reads only its first two untaken keys (best / second of the ratio test): every pending lane enters its best train into a
    // hashed table with an atomic min of the lane number; a lane whose two keys have no earlier lane in the table keeps both whatever
    // the earlier lanes decide (they can only take their own best), so its decision is the sequential loop's.  The lanes before the
    // first one that fails this test are decided at once, that one is then resolved alone (its keys may have shifted; it may need the
    // exact rescan), and the test is repeated on what is left.  Hash collisions only make the test stricter.
    const uint32_t k0 = kk[0];
    unsigned long long pend = __ballot(v && (k0 >> 16) <= 50u);   // best possible distance above TH_LOW: can never match
    while (pend) {
      const bool mine = (pend >> lane) & 1ull;
      uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
      int nun = 0;
      if (mine) {
#pragma unroll
        for (int j = 0; j < PS_BF_TOPK; j++) {
          const uint32_t key = kk[j];
          if (key != 0xFFFFFFFFu && nun < 2) {
            const uint32_t idx = key & 0xFFFF;
            if (!((taken[idx >> 5] >> (idx & 31)) & 1u)) { if (nun == 0) b1 = key; else b2 = key; nun++; }
          }
        }
      }
      const bool rescan = mine && nun < 2 && P.nt > PS_BF_TOPK;   // the list is exhausted: only the serial path can decide
      if (mine && b1 != 0xFFFFFFFFu) atomicMin(&claim[(b1 & 0xFFFF) & (BF_CLAIM - 1)], (uint32_t)lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      bool dirty = rescan;
      if (mine && b1 != 0xFFFFFFFFu) dirty = dirty || claim[(b1 & 0xFFFF) & (BF_CLAIM - 1)] < (uint32_t)lane;
      if (mine && b2 != 0xFFFFFFFFu) dirty = dirty || claim[(b2 & 0xFFFF) & (BF_CLAIM - 1)] < (uint32_t)lane;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (mine && b1 != 0xFFFFFFFFu) claim[(b1 & 0xFFFF) & (BF_CLAIM - 1)] = 0xFFFFFFFFu;
      const unsigned long long dm = __ballot(mine && dirty);
      const int fd = dm ? __ffsll((long long)dm) - 1 : 64;
      const unsigned long long cleanm = fd < 64 ? (pend & ((1ull << fd) - 1ull)) : pend;
      {
        const int d1 = b1 == 0xFFFFFFFFu ? 256 : (int)(b1 >> 16), d2 = b2 == 0xFFFFFFFFu ? 256 : (int)(b2 >> 16);
        const bool accept = ((cleanm >> lane) & 1ull) && d1 <= 50 && (float)d1 < __fmul_rn(nn_ratio, (float)d2);
        if (accept) {
          const int bi = (int)(b1 & 0xFFFF);
          atomicOr(&taken[bi >> 5], 1u << (bi & 31));
          out[bi] = q0 + lane;
          if (check_ori) {
            float rot = __fsub_rn(qa, tang_lds ? tang_s[bi] : tang[P.t_off + bi]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == 30) bin = 0;
            atomicAdd(&hist[bin], 1);
            bin_of[bi] = (uint8_t)bin;
          }
        }
        nm += __popcll(__ballot(accept));
      }
      pend &= ~cleanm;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (fd < 64) {
        // ---- the first query that shares a train with an earlier one: alone, on the bitmap as the lanes before it left it ----
        const int b = fd;
        pend &= ~(1ull << b);
        const int q = q0 + b;
        uint32_t key = 0xFFFFFFFFu;                // lane j < 8: key j of query q (out of lane b's registers)
#pragma unroll
        for (int j = 0; j < PS_BF_TOPK; j++) { const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)kk[j], b); if (lane == j) key = kj; }
        const uint32_t idx = key & 0xFFFF;
        const bool untaken = key != 0xFFFFFFFFu && !((taken[idx >> 5] >> (idx & 31)) & 1u);
        const unsigned long long um = __ballot(untaken);
        uint32_t best, second;
        if (__popcll(um) >= 2 || P.nt <= PS_BF_TOPK) {
          const int f = um ? __ffsll((long long)um) - 1 : -1;
          const unsigned long long um2 = um & (um - 1);
          const int s2 = um2 ? __ffsll((long long)um2) - 1 : -1;
          best = f >= 0 ? (uint32_t)__shfl((int)key, f) : (256u << 16);
          second = s2 >= 0 ? (uint32_t)__shfl((int)key, s2) : (256u << 16);
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
          const float qangle = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qa), b));
          if (lane == 0) {
            taken[bi >> 5] |= 1u << (bi & 31);
            out[bi] = q;
            if (check_ori) {
              float rot = __fsub_rn(qangle, tang_lds ? tang_s[bi] : tang[P.t_off + bi]);
              if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
              int bin = (int)roundf(__fmul_rn(rot, factor));
              if (bin == 30) bin = 0;
              hist[bin]++;
              bin_of[bi] = (uint8_t)bin;
            }
          }
          nm++;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
  if (lane == 0) nmatch[pbi] = nm;
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
  if (nblocks > 0) {
    hipLaunchKernelGGL((bf_topk_small<256, -1>), dim3(nblocks), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk, (const int32_t*)nullptr);
    hipLaunchKernelGGL((bf_topk_small<BF_SMALL_NT, 256>), dim3(nblocks), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk, (const int32_t*)nullptr);
    hipLaunchKernelGGL(bf_topk, dim3(nblocks), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk, (const int32_t*)nullptr);
  }
  hipLaunchKernelGGL(bf_resolve, dim3((nprob + 7) & ~7), dim3(64), 0, st, probs, qdesc, qang, qvalid, tdesc, tang, topk, out,
                     nmatch, nn_ratio, check_ori, nprob);
}
// the same with a block table that was built on the device: `d_count` entries, `grid` workgroups loop over them
extern "C" void psk_bf_launch_dev(const BfBlock* blocks, const int32_t* d_count, int grid, const BfProb* probs, int nprob, const uint8_t* qdesc,
                                  const float* qang, const uint8_t* qvalid, const uint8_t* tdesc, const float* tang, uint32_t* topk, int32_t* out,
                                  int32_t* nmatch, float nn_ratio, int check_ori, hipStream_t st) {
  hipLaunchKernelGGL((bf_topk_small<256, -1>), dim3(grid), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk, d_count);
  // (215 registers: two workgroups per CU - a grid beyond 512 adds waiting workgroups, not parallelism)
  hipLaunchKernelGGL((bf_topk_small<BF_SMALL_NT, 256>), dim3(grid < 512 ? grid : 512), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk, d_count);
  // (the launch for the problems with more than BF_SMALL_NT trains - none in most steps, the workgroups then return at once - with 32
  // workgroups: each needs 64 KB of LDS before it can start, and 1024 of them waited 220 us for their turn beside the other
  // lockstep groups' kernels; 5 us alone)
  hipLaunchKernelGGL(bf_topk, dim3(grid < 32 ? grid : 32), dim3(TOPK_T), 0, st, blocks, probs, qdesc, tdesc, topk, d_count);
  hipLaunchKernelGGL(bf_resolve, dim3((nprob + 7) & ~7), dim3(64), 0, st, probs, qdesc, qang, qvalid, tdesc, tang, topk, out,
                     nmatch, nn_ratio, check_ori, nprob);
}
extern "C" void psk_hamming_matrix_launch(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out, hipStream_t st) {
  hipLaunchKernelGGL(hamming_matrix, dim3((nt + 255) / 256, nq), dim3(256), 0, st, q, nq, t, nt, out);
}

// ================================================================================================
// Windowed matching: the three ORBmatcher::SearchByProjection overloads
//   (Frame&, const Frame&, th, mono)                 /root/reference/src/ORBmatcher.cc:1613-1756
//   (Frame&, const vector<MapPoint*>&, th)           :68-155
//   (Frame&, nOrder, const vector<MapObjectPoint*>&, th)   :157-248
// with Frame::GetFeaturesInArea (/root/reference/src/Frame.cc:1808-1861) as the candidate generator.
//   pj_project  : frame-to-frame variant only — float projection of the last frame's map points
//   pj_gather   : four queries per wave — grid window walk in the reference's order (ix, iy, cell order), static
//                 filters (level, |dx|,|dy| < r, stereo uR gate, bbox), Hamming distance; candidates are
//                 stored in traversal order as keys  dist << 23 | position << IDXB | train index
//   pj_resolve  : one wave per problem — the order-dependent part: queries in order, trains blocked by an
//                 earlier assignment are skipped, best / second best by two wave-min reductions, ratio test,
//                 rotation histogram
// ================================================================================================
namespace {



__device__ __forceinline__ float dot3_f(const float* R, float x, float y, float z) {
  // cv::Mat(float) * cv::Mat(float): cv::gemm accumulates float products in double
  return (float)((double)R[0] * (double)x + (double)R[1] * (double)y + (double)R[2] * (double)z);
}

__global__ __launch_bounds__(256) void pj_project(PjArrays A) {
  const PjProb& P = A.prob[blockIdx.y];   // by reference: a private copy indexed by the octave (P.scale[oct]) would live in scratch memory
  if (!P.frame_mode) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P.nq) return;
  const int q = P.q_off + i;
  if (!A.qvalid[q]) return;
  // twc = -Rcw^T tcw ; tlc = Rlw twc + tlw  (ORBmatcher.cc:1624-1633)
  const float t0 = P.tcw[3], t1 = P.tcw[7], t2 = P.tcw[11];
  float twc[3];
  for (int r = 0; r < 3; r++)
    twc[r] = (float)(-((double)P.tcw[r] * (double)t0 + (double)P.tcw[4 + r] * (double)t1 + (double)P.tcw[8 + r] * (double)t2));
  const float tlc2 = __fadd_rn(dot3_f(&P.tlw[8], twc[0], twc[1], twc[2]), P.tlw[11]);
  const bool fwd = tlc2 > P.mb && !P.mono, bwd = -tlc2 > P.mb && !P.mono;
  const float X = A.qxw[3 * q], Y = A.qxw[3 * q + 1], Z = A.qxw[3 * q + 2];
  const float xc = __fadd_rn(dot3_f(&P.tcw[0], X, Y, Z), P.tcw[3]);
  const float yc = __fadd_rn(dot3_f(&P.tcw[4], X, Y, Z), P.tcw[7]);
  const float zc = __fadd_rn(dot3_f(&P.tcw[8], X, Y, Z), P.tcw[11]);
  const float invzc = (float)(1.0 / (double)zc);
  bool ok = !(invzc < 0);
  const float u = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, xc), invzc), P.cx);
  const float v = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, yc), invzc), P.cy);
  if (u < P.bounds[0] || u > P.bounds[1] || v < P.bounds[2] || v > P.bounds[3]) ok = false;
  const int oct = A.qoct[q];
  const float radius = __fmul_rn(P.th, P.scale[oct]);
  A.qu[q] = u; A.qv[q] = v;
  A.qur[q] = __fsub_rn(u, __fmul_rn(P.mbf, invzc));
  A.qrad[q] = radius; A.qrer[q] = radius;
  A.qminl[q] = fwd ? oct : (bwd ? 0 : oct - 1);
  A.qmaxl[q] = fwd ? -1 : (bwd ? oct : oct + 1);
  if (!ok) A.qvalid[q] = 0;
}

// reductions inside a 16-lane DPP row (every lane of the row gets the result)
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {
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
__device__ __forceinline__ int row_scan_inclusive(int v) {          // Kogge-Stone over the row (row_shr 1, 2, 4, 8; zero fill)
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  return v;
}

// Candidate keys are  dist << 23 | position << IDXB | train index  with 23 - IDXB position bits: IDXB = 15 is the everyday format
// (32 767 features per frame, 256 candidates per window); IDXB = 13 (8191 features, 1024 candidates) serves the problems whose
// windows overflowed the first one, so that a dense window costs a second pass instead of the result.
//
// FOUR queries per wave, one per 16-lane DPP row.  A window is a handful of grid columns, each one contiguous range of the
// cell-sorted feature list; a lane per column reads the range bounds, a row prefix sum numbers the window's candidates in the
// reference's traversal order (column by column), and then a lane per CANDIDATE finds its column (binary search over the row's
// prefix sums through ds_bpermute) and filters / measures it: bounds, feature index, feature fields, descriptor - four dependent
// memory round trips for a whole window, where a wave per query walking column after column paid four per column.
template <int IDXB>
__global__ __launch_bounds__(256) void pj_gather(PjArrays A, int nprob) {
  constexpr int CAP = 1 << (23 - IDXB);
  // r05: one problem per XCD at a time (workgroup b runs on XCD b % 8, every XCD has its own L2): the launch is (8 x blocks per problem,
  // ceil(problems / 8)) and problem = 8 y + x % 8, so the workgroups that walk one frame's grid, candidate records and descriptors
  // (~110 KB) share an L2 instead of fetching them eight times
  // (the XCD a problem lands on rotates with y: the object search has its live problems at p % 8 in {0, 1} - two detections per sequence of eight
  // slots - and would otherwise use two of the eight XCDs)
  const int pb = (int)blockIdx.y * 8 + (((int)blockIdx.x + (int)blockIdx.y) & 7), bxq = (int)blockIdx.x >> 3;
  if (pb >= nprob) return;
  const PjProb& P = A.prob[pb];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
  const int nq = P.nq;
  const int qi0 = bxq * 16 + wave * 4;
  if (qi0 >= nq) return;
  const int qi = qi0 + grp;
  const bool live = qi < nq;
  const int q = P.q_off + (live ? qi : nq - 1);
  const bool valid = live && A.qvalid[q];
  const float x = A.qu[q], y = A.qv[q], r = A.qrad[q], rer = A.qrer[q], ur = A.qur[q];
  const int minLevel = A.qminl[q], maxLevel = A.qmaxl[q];
  // Frame::GetFeaturesInArea cell range (Frame.cc:1813-1827)
  const float min_x = P.min_x, min_y = P.min_y, gw_inv = P.gw_inv, gh_inv = P.gh_inv;
  const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), gw_inv)));
  const int nMaxCellX = min(PS_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), gw_inv)));
  const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), gh_inv)));
  const int nMaxCellY = min(PS_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), gh_inv)));
  int count = 0;
  uint32_t mk[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // this lane's four smallest keys among the trains free at entry
  if (valid && nMinCellX < PS_GRID_COLS && nMaxCellX >= 0 && nMinCellY < PS_GRID_ROWS && nMaxCellY >= 0) {
    const bool check = (minLevel > 0) || (maxLevel >= 0);
    const uint4* qd = reinterpret_cast<const uint4*>(A.qdesc + (size_t)q * 32);
    const uint4 a0 = qd[0], a1 = qd[1];
    const int32_t* coff = A.cell_off + P.grid_off;
    const int t_off = P.t_off;
    const bool use_bbox = P.use_bbox != 0;
    const int rowbase = (lane & 48) << 2;                       // ds_bpermute byte address of the row's lane 0
    for (int cx0 = nMinCellX; cx0 <= nMaxCellX; cx0 += 16) {    // 16 columns at a time (a window rarely has more)
      const int ix = cx0 + l16;
      int cb = 0, n = 0;
      if (ix <= nMaxCellX) { cb = coff[ix * PS_GRID_ROWS + nMinCellY]; n = coff[ix * PS_GRID_ROWS + nMaxCellY + 1] - cb; }
      const int excl = row_scan_inclusive(n) - n;               // candidates in the columns before this one
      const int T = row_sum_i32(n);
      for (int m0 = 0; m0 < T; m0 += 16) {
        const int m = m0 + l16;                                 // traversal position of this lane's candidate
        // its column: the last one whose prefix is <= m (empty columns share the prefix of their successor and lose to it)
        int c = 0;
#pragma unroll
        for (int step = 8; step >= 1; step >>= 1) {
          const int cc = c + step;
          const int pv = __builtin_amdgcn_ds_bpermute(rowbase + 4 * min(cc, 15), excl);
          if (cc < 16 && pv <= m) c = cc;
        }
        const int cexcl = __builtin_amdgcn_ds_bpermute(rowbase + 4 * c, excl), cbase = __builtin_amdgcn_ds_bpermute(rowbase + 4 * c, cb);
        bool pass = false;
        int j = 0, dist = 0;
        if (m < T) {
          j = A.cell_idx[t_off + cbase + (m - cexcl)];
          const int t = t_off + j;
          const int oc = A.toct[t];
          pass = true;
          if (check) {
            if (oc < minLevel) pass = false;
            if (maxLevel >= 0 && oc > maxLevel) pass = false;
          }
          const float dx = __fsub_rn(A.tx[t], x), dy = __fsub_rn(A.ty[t], y);
          if (!(fabsf(dx) < r && fabsf(dy) < r)) pass = false;
          if (pass && use_bbox && !A.tbbox[t]) pass = false;
          if (pass) {
            const float tu = A.tur[t];
            if (tu > 0.f && fabsf(__fsub_rn(ur, tu)) > rer) pass = false;
          }
          if (pass) {
            const uint4* td = reinterpret_cast<const uint4*>(A.tdesc + (size_t)t * 32);
            dist = hamming256(a0, a1, td[0], td[1]);
          }
        }
        const uint32_t mrow = (uint32_t)(__ballot(pass) >> (lane & 48)) & 0xFFFFu;     // the row's passing lanes
        const int pos = count + __popc(mrow & ((1u << l16) - 1u));
        if (pass && pos < CAP) {
          const uint32_t key = ((uint32_t)dist << 23) | ((uint32_t)pos << IDXB) | (uint32_t)j;
          A.cand[((size_t)P.c_off + qi) * CAP + pos] = key;
          if (!A.tocc[t_off + j]) {   // sorted insertion
            uint32_t kk = key;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) { const uint32_t lo = min(mk[rr], kk); kk = max(mk[rr], kk); mk[rr] = lo; }
          }
        }
        count += __popc(mrow);
      }
    }
  }
  // the four smallest keys of the query under the occupancy at entry (keys are unique: they carry the traversal position):
  // pj_resolve takes the first ones that no earlier query of the same call has claimed and only goes back to the candidate
  // list when all four are gone
  uint32_t top[4];
  uint32_t prev = 0;
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    uint32_t c = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 3; u >= 0; u--) if (rr == 0 || mk[u] > prev) c = min(c, mk[u]);
    top[rr] = row_min_u32(c);
    prev = top[rr];
  }
  if (l16 == 0 && live) {
    if (count > CAP) { atomicAdd(&A.overflow[pb], 1); count = CAP; }
    A.ncand[q] = count;
    A.ttop[q] = make_uint4(top[0], top[1], top[2], top[3]);
  }
}

#ifndef PJ_TBL
#define PJ_TBL 4096   // (8192: 3 % fewer lanes lose the parallel path to a hash collision, but 16 KB more LDS per problem - measured slower)
#endif
// (r06) Launched with 64 threads per problem in a throughput-sized batch, with 256 when the call holds a few problems (one sequence per
// handle: BASELINE configs[4]): the order-dependent loop is wave 0's either way, the set-up (occupancy bitmap, claim table, match array,
// octave copy) and the rotation-histogram tail are spread over all the threads there are - a third of the kernel's 115 us at one problem.
template <int IDXB>
__global__ __launch_bounds__(256) void pj_resolve(PjArrays A, int nprob_total) {
  constexpr int CAP = 1 << (23 - IDXB);
  constexpr uint32_t IDXM = (1u << IDXB) - 1u;
  __shared__ uint32_t blocked[1024];   // up to 32768 train features
  __shared__ int hist[32];
  __shared__ uint32_t newly[1024];     // trains blocked by this call
  extern __shared__ uint8_t loct[];    // train octaves (ratio test), staged once: no global load inside the serial loop; sized by the launch
                                       // for the call's largest train set (with room for all 32768 the workgroup took 72 KB: two per CU)
  __shared__ uint32_t first_lane[PJ_TBL];   // hashed train -> earliest lane of the current block that lists it among its four keys
  __shared__ int s_removed, s_nm;
  const int nth = (int)blockDim.x;
  const int pbi = ((int)blockIdx.x & ~7) | (((int)blockIdx.x + ((int)blockIdx.x >> 3)) & 7);      // (rotated inside every group of eight: see bf_resolve)
  if (pbi >= nprob_total) return;
  const PjProb P = A.prob[pbi];
  const int lane = threadIdx.x;
  int32_t* match = A.match + P.t_off;
  if (lane == 0) { s_removed = 0; s_nm = 0; }
  for (int w = lane; w < (P.nt + 31) / 32; w += nth) {
    uint32_t bits = 0;
    for (int b = 0; b < 32; b++) {
      const int j = w * 32 + b;
      if (j < P.nt && A.tocc[P.t_off + j]) bits |= 1u << b;
    }
    blocked[w] = bits;
    newly[w] = 0;
  }
  if (lane < 32) hist[lane] = 0;
  for (int w = lane; w < PJ_TBL; w += nth) first_lane[w] = 0xFFFFFFFFu;
  for (int j = lane; j < P.nt; j += nth) match[j] = -1;
  if (P.ratio_test)
    for (int j = lane; j < P.nt; j += nth) loct[j] = (uint8_t)A.toct[P.t_off + j];
  __syncthreads();
  const float factor = 30 / 360.0f;
  int nm = 0;
#ifdef PS_PJ_PROFILE
  const long long pj_t0 = wall_clock64();
  long long pj_t1 = 0;
  int pj_clean = 0, pj_dirty = 0, pj_serial_blocks = 0;
#endif
  if (lane < 64) {       // ---- the order-dependent part: wave 0 ----
  // Queries are taken in order (the assignment is order dependent), 64 at a time: the lane-resident candidate counts give
  // the non-empty queries of the block as a bit mask, and the first 64 candidate keys of the NEXT non-empty query are
  // requested before the current one is reduced, so the global-memory latency is paid once per block, not once per query.
  // newly: bitmap of the trains blocked by THIS call (the occupancy at entry is already excluded from tbest / tsecond)
  const uint4 no_keys = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  int ncv_next = lane < P.nq ? A.ncand[P.q_off + lane] : 0;
  int obsv_next = lane < P.nq ? (int)A.qobs[P.q_off + lane] : 0;
  uint4 tt_next = lane < P.nq ? A.ttop[P.q_off + lane] : no_keys;
  for (int q0 = 0; q0 < P.nq; q0 += 64) {
    const bool qin = q0 + lane < P.nq;
    const int qq = P.q_off + q0 + lane;
    const int ncv = ncv_next, obsv = obsv_next;
    const uint4 tt = tt_next;
    {   // the next block's per-query data is requested now and arrives while this block is decided
      const bool nin = q0 + 64 + lane < P.nq;
      ncv_next = nin ? A.ncand[qq + 64] : 0;
      obsv_next = nin ? (int)A.qobs[qq + 64] : 0;
      tt_next = nin ? A.ttop[qq + 64] : no_keys;
    }
    const uint32_t tk[4] = {tt.x, tt.y, tt.z, tt.w};
    if (qin) A.qbest[qq] = -1;
    // claims of earlier blocks against the four keys: one LDS lookup per key and lane, in parallel
    int stale = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (tk[r] != 0xFFFFFFFFu) { const uint32_t j = tk[r] & IDXM; stale |= (int)((newly[j >> 5] >> (j & 31)) & 1u) << r; }
    unsigned long long pend = __builtin_amdgcn_ballot_w64(ncv > 0);
    // ---- queries whose outcome cannot depend on the other queries of the block are decided by all lanes at once ----
    // A query's choice among its four keys only changes when an EARLIER query of the block claims one of those trains.  Every
    // lane enters its (up to four) trains into a hashed table with an atomic min of the lane number; a lane that is the
    // earliest in all of its buckets shares no train with any earlier lane (hash collisions only make the test stricter), so
    // its decision is the one the sequential loop would take, whatever the others do.  The rest goes through the loop below.
    // A query that could run out of unclaimed keys (fewer exclusive ones than the decision reads, and a candidate list longer
    // than the four) might fall back to scanning its whole list, which can pick any train: it stays in the sequential loop ...
    {
      const int want = P.ratio_test ? 2 : 1;
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (ncv > 0 && tk[r] != 0xFFFFFFFFu) atomicMin(&first_lane[(tk[r] & IDXM) & (PJ_TBL - 1)], (uint32_t)lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
      int found = 0, exclusive = 0;
      bool clean = ncv > 0;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (ncv > 0 && tk[r] != 0xFFFFFFFFu) {
          const bool mine = first_lane[(tk[r] & IDXM) & (PJ_TBL - 1)] == (uint32_t)lane;
          clean = clean && mine;
          if (!((stale >> r) & 1)) {
            if (found == 0) best = tk[r]; else if (found == 1) second = tk[r];
            found++;
            exclusive += mine ? 1 : 0;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (ncv > 0 && tk[r] != 0xFFFFFFFFu) first_lane[(tk[r] & IDXM) & (PJ_TBL - 1)] = 0xFFFFFFFFu;
      // (fewer than four keys: they are ALL the candidates that were free at entry, a scan cannot find another one)
      const bool may_scan = ncv > 4 && tk[3] != 0xFFFFFFFFu && exclusive < want;
      // ... and so does everything after it: only the lanes before the first such query take the parallel path
      const unsigned long long scanners = __builtin_amdgcn_ballot_w64(may_scan);
      const unsigned long long before = scanners ? ((1ull << (__ffsll((long long)scanners) - 1)) - 1ull) : ~0ull;
      clean = clean && ((before >> lane) & 1ull);
      {
        bool accept = clean && best != 0xFFFFFFFFu && (int)(best >> 23) <= P.th_dist;
        const int bestIdx = (int)(best & IDXM);
        if (accept && P.ratio_test && second != 0xFFFFFFFFu) {
          const int l1 = loct[bestIdx], l2 = loct[second & IDXM];
          if (l1 == l2 && (float)(int)(best >> 23) > __fmul_rn(P.nn_ratio, (float)(int)(second >> 23))) accept = false;
        }
        if (accept) {
          if (obsv) atomicOr(&newly[bestIdx >> 5], 1u << (bestIdx & 31));
          atomicMax(&match[bestIdx], q0 + lane);
          A.qbest[qq] = bestIdx;
        }
        nm += __popcll(__builtin_amdgcn_ballot_w64(accept));
        pend &= ~__builtin_amdgcn_ballot_w64(clean);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the remaining queries see the claims just made by earlier lanes of the block (later lanes share no train with them)
        stale = 0;
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (tk[r] != 0xFFFFFFFFu) { const uint32_t j = tk[r] & IDXM; stale |= (int)((newly[j >> 5] >> (j & 31)) & 1u) << r; }
#ifdef PS_PJ_PROFILE
        pj_clean += __popcll(__builtin_amdgcn_ballot_w64(clean)); pj_dirty += __popcll(pend); pj_serial_blocks += scanners ? 1 : 0;
#endif
      }
    }
    int accidx = -1;        // lane k: the train taken by the k-th observed acceptance of this block
    int nacc = 0;
    int allq = -1, allt = 0, nall = 0;   // lane k: query and train of the k-th acceptance of this block (at most 64)
    while (pend) {
      const int i = __ffsll((long long)pend) - 1;
      pend &= pend - 1;
      const int qi = q0 + i;
      const int nc = __builtin_amdgcn_readlane(ncv, i);
      const int sb = __builtin_amdgcn_readlane(stale, i);
      // first (and, for the ratio test, second) of the four keys that nobody has claimed
      uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
      bool need_scan = false;
      {
        int found = 0;
        const int want = P.ratio_test ? 2 : 1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (found < want) {
            const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)tk[r], i);
            if (key != 0xFFFFFFFFu) {
              bool taken = (sb >> r) & 1;
              if (!taken && nacc > 0) taken = __builtin_amdgcn_ballot_w64(accidx == (int)(key & IDXM)) != 0ull;
              if (!taken) { if (found == 0) best = key; else second = key; found++; }
            }
          }
        }
        need_scan = found < want && nc > 4;   // the list may hold unclaimed candidates beyond the four
      }
      if (need_scan) {
        uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
        for (int c = lane; c < nc; c += 64) {
          const uint32_t k = A.cand[((size_t)P.c_off + qi) * CAP + c];
          const uint32_t j = k & IDXM;
          if (((blocked[j >> 5] | newly[j >> 5]) >> (j & 31)) & 1u) continue;
          if (k < m1) { m2 = m1; m1 = k; } else if (k < m2) m2 = k;
        }
        best = wave_min_u32(m1);
        second = wave_min_u32(m1 == best ? m2 : m1);
      }
      if (best == 0xFFFFFFFFu) continue;
      const int bestDist = (int)(best >> 23), bestIdx = (int)(best & IDXM);
      if (bestDist > P.th_dist) continue;
      if (P.ratio_test && second != 0xFFFFFFFFu) {
        const int d2 = (int)(second >> 23);
        const int l1 = loct[bestIdx], l2 = loct[second & IDXM];
        if (l1 == l2 && (float)bestDist > __fmul_rn(P.nn_ratio, (float)d2)) continue;
        // (no second candidate: bestLevel2 = -1 never equals an octave -> accepted)
      }
      const int observed = __builtin_amdgcn_readlane(obsv, i);
      // nothing leaves the wave inside the serial loop (a release fence after a global store waits for the store: ~0.4 us per
      // accepted query): the claim goes to the LDS bitmap and to the lane-resident lists, the outputs are written per block
      if (observed && lane == 0) atomicOr(&newly[bestIdx >> 5], 1u << (bestIdx & 31));
      if (lane == nall) { allq = qi; allt = bestIdx; }
      nall++;
      if (observed) {
        if (lane == nacc) accidx = bestIdx;
        nacc++;
      }
      nm++;
      __builtin_amdgcn_wave_barrier();
    }
    // outputs of the block: the reference's "last assignment wins" for a keypoint claimed twice (possible when the claimant
    // is not observed) is the larger query index, i.e. an atomic max
    if (lane < nall) {
      atomicMax(&match[allt], allq);
      A.qbest[P.q_off + allq] = allt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (lane == 0) s_nm = nm;
  }                      // ---- wave 0 ----
  __syncthreads();
  nm = s_nm;
#ifdef PS_PJ_PROFILE
  pj_t1 = wall_clock64();
#endif
  if (P.check_ori) {
    // rotHist[bin].push_back(...) of every assignment (ORBmatcher.cc:1716-1726); order inside a bin is irrelevant
    for (int qi = lane; qi < P.nq; qi += nth) {
      const int q = P.q_off + qi;
      const int bi = A.qbest[q];
      if (bi >= 0) {
        float rot = __fsub_rn(A.qang[q], A.tang[P.t_off + bi]);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, factor));
        if (bin == 30) bin = 0;
        atomicAdd(&hist[bin], 1);
        A.qbin[q] = (uint8_t)bin;
      }
    }
    __syncthreads();
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
    for (int qi = lane; qi < P.nq; qi += nth) {
      const int q = P.q_off + qi;
      const int bi = A.qbest[q];
      if (bi >= 0) {
        const int b = A.qbin[q];
        if (b != i1 && b != i2 && b != i3) { match[bi] = -2; removed++; }   // -2: assigned in this call, then set to NULL by the rotation check
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) removed += __shfl_xor(removed, d);
    if ((lane & 63) == 0 && removed) atomicAdd(&s_removed, removed);
    __syncthreads();
    nm -= s_removed;
  }
  if (lane == 0) A.nmatch[pbi] = nm;
#ifdef PS_PJ_PROFILE
  if (lane == 0 && blockIdx.x == 0) printf("pj_resolve nq %d nt %d: loop %lld ticks, tail %lld ticks, matches %d, decided in parallel %d, sequentially %d, blocks with a scanning query %d\n", P.nq, P.nt, pj_t1 - pj_t0, wall_clock64() - pj_t1, nm, pj_clean, pj_dirty, pj_serial_blocks);
#endif
}

}  // namespace

// max_nt: an upper bound of the problems' train counts (at most 32768)
extern "C" void psk_pj_launch(const PjArrays* arrays, int nprob, int max_nq, int max_nt, int any_frame_mode, int wide, hipStream_t st) {
  const PjArrays A = *arrays;
  const size_t lds = (size_t)((max_nt < 1 ? 1 : max_nt > 32768 ? 32768 : max_nt) + 15) & ~(size_t)15;
  const int rt = nprob <= 64 ? 256 : 64;       // pj_resolve's threads per problem: see the kernel
  if (any_frame_mode) hipLaunchKernelGGL(pj_project, dim3((max_nq + 255) / 256, nprob), dim3(256), 0, st, A);
  if (wide) {
    hipLaunchKernelGGL(pj_gather<13>, dim3(8 * ((max_nq + 15) / 16), (nprob + 7) / 8), dim3(256), 0, st, A, nprob);
    hipLaunchKernelGGL(pj_resolve<13>, dim3((nprob + 7) & ~7), dim3(rt), lds, st, A, nprob);
  } else {
    hipLaunchKernelGGL(pj_gather<15>, dim3(8 * ((max_nq + 15) / 16), (nprob + 7) / 8), dim3(256), 0, st, A, nprob);
    hipLaunchKernelGGL(pj_resolve<15>, dim3((nprob + 7) & ~7), dim3(rt), lds, st, A, nprob);
  }
}

// ================================================================================================
// MapPoint / MapObjectPoint::ComputeDistinctiveDescriptors (/root/reference/src/MapObjectPoint.cc:379-436, MapPoint.cc:366;
// SURVEY.md 8f-4): the observation whose descriptor has the least median Hamming distance to the others.  One wave per
// point: lane = row of the distance matrix (kept in LDS), median by rank counting, smallest (median, index) by a wave-min.
// ================================================================================================
namespace {
#define DD_MAX 128   // observations per point
__global__ __launch_bounds__(64) void distinctive_desc(const uint8_t* desc, const int32_t* off, int32_t* best) {
  __shared__ uint16_t rows[64][DD_MAX + 2];
  const int p = blockIdx.x, lane = threadIdx.x;
  const int o = off[p], N = off[p + 1] - o;
  if (N <= 0) { if (lane == 0) best[p] = -1; return; }
  const uint4* D = reinterpret_cast<const uint4*>(desc + (size_t)o * 32);
  const int k = (int)(0.5 * (N - 1));   // vDists[0.5*(N-1)]
  uint32_t bestkey = 0xFFFFFFFFu;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    if (i < N) {
      const uint4 a0 = D[2 * i], a1 = D[2 * i + 1];
      for (int j = 0; j < N; j++) rows[lane][j] = (uint16_t)hamming256(a0, a1, D[2 * j], D[2 * j + 1]);
      int median = 0;
      for (int j = 0; j < N; j++) {
        const int dj = rows[lane][j];
        int rank = 0;
        for (int m = 0; m < N; m++) {
          const int dm = rows[lane][m];
          rank += (dm < dj || (dm == dj && m < j)) ? 1 : 0;
        }
        if (rank == k) median = dj;
      }
      bestkey = min(bestkey, ((uint32_t)median << 16) | (uint32_t)i);
    }
  }
  bestkey = wave_min_u32(bestkey);
  if (lane == 0) best[p] = (int32_t)(bestkey & 0xFFFF);
}
}  // namespace
extern "C" void psk_distinctive_launch(const uint8_t* desc, const int32_t* off, int32_t* best, int npoints, hipStream_t st) {
  hipLaunchKernelGGL(distinctive_desc, dim3(npoints), dim3(64), 0, st, desc, off, best);
}

// ================================================================================================
// The search half of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (/root/reference/src/ORBmatcher.cc:982-1136) and
// Fuse(ObjectKeyFrame*, vpMapObjectPoints, th) (:1138-1260), SURVEY.md 8f-4.  One wave per candidate point: the float
// projection and the gates are evaluated by every lane (they are wave-uniform), then the lanes walk the grid window of
// KeyFrame::GetFeaturesInArea in the reference's order; "first strict minimum" = wave-min of (distance, traversal position).
// ================================================================================================
namespace {
__global__ __launch_bounds__(256) void fuse_search(FuArrays A) {
  const FuProb& P = A.prob[blockIdx.y];   // by reference (see pj_project)
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (qi >= P.nq) return;
  const int q = P.q_off + qi;
  int out_idx = -1, out_dist = 256;
  bool ok = A.qvalid[q] != 0;
  const float X = A.qpos[3 * q], Y = A.qpos[3 * q + 1], Z = A.qpos[3 * q + 2];
  const float xc = __fadd_rn(dot3_f(&P.R[0], X, Y, Z), P.t[0]);
  const float yc = __fadd_rn(dot3_f(&P.R[3], X, Y, Z), P.t[1]);
  const float zc = __fadd_rn(dot3_f(&P.R[6], X, Y, Z), P.t[2]);
  if (zc < 0.0f) ok = false;
  const float invz = __fdiv_rn(1.0f, zc);
  const float u = __fadd_rn(__fmul_rn(P.fx, __fmul_rn(xc, invz)), P.cx);
  const float v = __fadd_rn(__fmul_rn(P.fy, __fmul_rn(yc, invz)), P.cy);
  const float ur = __fsub_rn(u, __fmul_rn(P.bf, invz));
  if (!((double)u >= P.bounds[0] && (double)u < P.bounds[1] && (double)v >= P.bounds[2] && (double)v < P.bounds[3])) ok = false;
  const float p0 = __fsub_rn(X, P.ow[0]), p1 = __fsub_rn(Y, P.ow[1]), p2 = __fsub_rn(Z, P.ow[2]);
  const float dist3D = (float)sqrt(__dadd_rn(__dadd_rn(__dmul_rn((double)p0, (double)p0), __dmul_rn((double)p1, (double)p1)), __dmul_rn((double)p2, (double)p2)));
  const float maxd = A.qmax[q], mind = A.qmin[q];
  if (dist3D < __fmul_rn(0.8f, mind) || dist3D > __fmul_rn(1.2f, maxd)) ok = false;
  const double dotn = __dadd_rn(__dadd_rn(__dmul_rn((double)p0, (double)A.qnormal[3 * q]), __dmul_rn((double)p1, (double)A.qnormal[3 * q + 1])),
                                __dmul_rn((double)p2, (double)A.qnormal[3 * q + 2]));
  if (dotn < __dmul_rn(0.5, (double)dist3D)) ok = false;
  if (ok) {
    const float ratio = __fdiv_rn(maxd, dist3D);
    int lvl = (int)ceil(__ddiv_rn(log((double)ratio), (double)P.log_scale));
    lvl = lvl < 0 ? 0 : (lvl >= P.n_levels ? P.n_levels - 1 : lvl);
    const float r = __fmul_rn(P.th, P.scale[lvl]);
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, P.min_x), r), P.gw_inv)));
    const int nMaxCellX = min(PS_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, P.min_x), r), P.gw_inv)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, P.min_y), r), P.gh_inv)));
    const int nMaxCellY = min(PS_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, P.min_y), r), P.gh_inv)));
    uint32_t best = 0xFFFFFFFFu;   // distance << 20 | traversal position (unique per candidate)
    int bestj = -1;
    if (nMinCellX < PS_GRID_COLS && nMaxCellX >= 0 && nMinCellY < PS_GRID_ROWS && nMaxCellY >= 0) {
      const uint4* qd = reinterpret_cast<const uint4*>(A.qdesc + (size_t)q * 32);
      const uint4 a0 = qd[0], a1 = qd[1];
      const int32_t* coff = A.cell_off + P.grid_off;
      int scanned = 0;
      for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        const int b = coff[ix * PS_GRID_ROWS + nMinCellY], e = coff[ix * PS_GRID_ROWS + nMaxCellY + 1];
        for (int k0 = b; k0 < e; k0 += 64) {
          const int k = k0 + lane;
          if (k < e) {
            const int j = A.cell_idx[P.t_off + k];
            const int t = P.t_off + j;
            const float ex = __fsub_rn(u, A.tx[t]), ey = __fsub_rn(v, A.ty[t]);
            const int oc = A.toct[t];
            bool pass = fabsf(ex) < r && fabsf(ey) < r;            // KeyFrame::GetFeaturesInArea
            if (oc < lvl - 1 || oc > lvl) pass = false;
            const float tu = A.tur[t];
            const float is2 = P.inv_sigma2[oc & 7];
            const float e2m = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
            if (tu >= 0.f) {
              const float er = __fsub_rn(ur, tu);
              if ((double)__fmul_rn(__fadd_rn(e2m, __fmul_rn(er, er)), is2) > 7.8) pass = false;
            } else {
              if ((double)__fmul_rn(e2m, is2) > 5.99) pass = false;
            }
            if (pass) {
              const uint4* td = reinterpret_cast<const uint4*>(A.tdesc + (size_t)t * 32);
              const uint32_t dist = (uint32_t)hamming256(a0, a1, td[0], td[1]);
              const uint32_t key = (dist << 20) | (uint32_t)min(scanned + (k - k0), 0xFFFFF);
              if (key < best) { best = key; bestj = j; }
            }
          }
          scanned += 64;
        }
      }
    }
    const uint32_t bk = wave_min_u32(best);
    if (bk != 0xFFFFFFFFu) {
      out_dist = (int)(bk >> 20);
      const unsigned long long holder = __builtin_amdgcn_ballot_w64(best == bk);
      const int src = __ffsll((long long)holder) - 1;
      const int bj = __builtin_amdgcn_readlane(bestj, src);
      if (out_dist <= 50) out_idx = bj;                              // TH_LOW
    }
  }
  if (lane == 0) { A.best_idx[q] = out_idx; A.best_dist[q] = out_dist; }
}
}  // namespace
extern "C" void psk_fuse_launch(const FuArrays* arrays, int nprob, int max_nq, hipStream_t st) {
  hipLaunchKernelGGL(fuse_search, dim3((max_nq + 3) / 4, nprob), dim3(256), 0, st, *arrays);
}
