// Host side of the matcher C-ABI (include/pointslot_hip.h): packs the caller's per-object problems into
// one upload, launches the kernels of match_kernels.hip on the handle's stream, copies the results back.
// Replaces ORBmatcher::SearchByBruceMatching / DescriptorDistance — /root/reference/src/ORBmatcher.cc.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "match_plan.h"
#include "ps_common.h"

extern "C" {
void psk_bf_launch(const BfBlock*, int, const BfProb*, int, const uint8_t*, const float*, const uint8_t*, const uint8_t*,
                   const float*, uint32_t*, int32_t*, int32_t*, float, int, hipStream_t);
void psk_hamming_matrix_launch(const uint8_t*, int, const uint8_t*, int, uint16_t*, hipStream_t);
}

struct ps_matcher {
  int device = 0;
  hipStream_t stream = nullptr;
  uint8_t* d_buf = nullptr;   // one growable device arena
  size_t d_bytes = 0;
  uint8_t* h_buf = nullptr;   // pinned staging of the same size
  size_t h_bytes = 0;
};

namespace {
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

int ensure(ps_matcher* m, size_t bytes) {
  if (bytes > m->d_bytes) {
    if (m->d_buf) hipFree(m->d_buf);
    m->d_buf = nullptr;
    PS_HIP(hipMalloc(&m->d_buf, bytes));
    m->d_bytes = bytes;
  }
  if (bytes > m->h_bytes) {
    if (m->h_buf) hipHostFree(m->h_buf);
    m->h_buf = nullptr;
    PS_HIP(hipHostMalloc(&m->h_buf, bytes, hipHostMallocDefault));
    m->h_bytes = bytes;
  }
  return PS_OK;
}
}  // namespace

extern "C" {

int ps_matcher_create(int device, ps_matcher** out) {
  if (!out) return ps_set_error(PS_ERR_INVALID, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ps_set_error(PS_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return ps_set_error(PS_ERR_INVALID, "bad device ordinal");
  PS_HIP(hipSetDevice(device));
  ps_matcher* m = new ps_matcher();
  m->device = device;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete m; return ps_set_error(PS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
  *out = m;
  return PS_OK;
}

void ps_matcher_destroy(ps_matcher* m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->stream) { hipStreamSynchronize(m->stream); hipStreamDestroy(m->stream); }
  if (m->d_buf) hipFree(m->d_buf);
  if (m->h_buf) hipHostFree(m->h_buf);
  delete m;
}

int ps_hamming_matrix(ps_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
  if (!m || !q || !t || !out || nq < 1 || nt < 1) return ps_set_error(PS_ERR_INVALID, "ps_hamming_matrix: bad argument");
  PS_HIP(hipSetDevice(m->device));
  const size_t oq = 0, ot = al((size_t)nq * 32), oo = ot + al((size_t)nt * 32), total = oo + al((size_t)nq * nt * 2);
  int rc = ensure(m, total);
  if (rc != PS_OK) return rc;
  memcpy(m->h_buf + oq, q, (size_t)nq * 32);
  memcpy(m->h_buf + ot, t, (size_t)nt * 32);
  PS_HIP(hipMemcpyAsync(m->d_buf, m->h_buf, oo, hipMemcpyHostToDevice, m->stream));
  psk_hamming_matrix_launch(m->d_buf + oq, nq, m->d_buf + ot, nt, (uint16_t*)(m->d_buf + oo), m->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(m->h_buf + oo, m->d_buf + oo, (size_t)nq * nt * 2, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  memcpy(out, m->h_buf + oo, (size_t)nq * nt * 2);
  return PS_OK;
}

int ps_match_bruteforce(ps_matcher* m, ps_bf_problem* probs, int nprob, float nn_ratio, int check_orientation) {
  if (!m || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_match_bruteforce: bad argument");
  PS_HIP(hipSetDevice(m->device));
  size_t tq = 0, tt = 0;
  std::vector<BfProb> dp(nprob);
  std::vector<BfBlock> blocks;
  for (int p = 0; p < nprob; p++) {
    ps_bf_problem& P = probs[p];
    if (P.nq < 0 || P.nt < 0 || P.nt > PS_BF_MAX_TRAIN || (P.nq > 0 && (!P.q_desc || !P.q_angle || !P.q_valid)) ||
        (P.nt > 0 && (!P.t_desc || !P.t_angle || !P.query_of_train)))
      return ps_set_error(PS_ERR_INVALID, "brute-force problem %d: bad sizes or null pointers (nt <= %d)", p, PS_BF_MAX_TRAIN);
    dp[p] = BfProb{(int32_t)tq, P.nq, (int32_t)tt, P.nt};
    if (P.nt > 0)
      for (int q0 = 0; q0 < P.nq; q0 += PS_BF_QPB)
        blocks.push_back(BfBlock{p, q0, P.nq - q0 < PS_BF_QPB ? P.nq - q0 : PS_BF_QPB, 0});
    tq += P.nq;
    tt += P.nt;
  }
  // arena layout (host staging mirrors the device arena)
  size_t off = 0;
  const size_t o_prob = off; off += al(sizeof(BfProb) * nprob);
  const size_t o_blk = off;  off += al(sizeof(BfBlock) * (blocks.size() + 1));
  const size_t o_qd = off;   off += al(tq * 32 + 32);
  const size_t o_qa = off;   off += al(tq * 4 + 4);
  const size_t o_qv = off;   off += al(tq + 1);
  const size_t o_td = off;   off += al(tt * 32 + 32);
  const size_t o_ta = off;   off += al(tt * 4 + 4);
  const size_t in_bytes = off;
  const size_t o_out = off;  off += al(tt * 4 + 4);
  const size_t o_nm = off;   off += al((size_t)nprob * 4);
  const size_t out_bytes = off - o_out;
  const size_t o_topk = off; off += al(tq * PS_BF_TOPK * 4 + 4);
  int rc = ensure(m, off);
  if (rc != PS_OK) return rc;
  uint8_t* H = m->h_buf;
  memcpy(H + o_prob, dp.data(), sizeof(BfProb) * nprob);
  if (!blocks.empty()) memcpy(H + o_blk, blocks.data(), sizeof(BfBlock) * blocks.size());
  for (int p = 0; p < nprob; p++) {
    const ps_bf_problem& P = probs[p];
    if (P.nq > 0) {
      memcpy(H + o_qd + (size_t)dp[p].q_off * 32, P.q_desc, (size_t)P.nq * 32);
      memcpy(H + o_qa + (size_t)dp[p].q_off * 4, P.q_angle, (size_t)P.nq * 4);
      memcpy(H + o_qv + (size_t)dp[p].q_off, P.q_valid, (size_t)P.nq);
    }
    if (P.nt > 0) {
      memcpy(H + o_td + (size_t)dp[p].t_off * 32, P.t_desc, (size_t)P.nt * 32);
      memcpy(H + o_ta + (size_t)dp[p].t_off * 4, P.t_angle, (size_t)P.nt * 4);
    }
  }
  uint8_t* D = m->d_buf;
  PS_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, m->stream));
  psk_bf_launch((const BfBlock*)(D + o_blk), (int)blocks.size(), (const BfProb*)(D + o_prob), nprob, D + o_qd,
                (const float*)(D + o_qa), D + o_qv, D + o_td, (const float*)(D + o_ta), (uint32_t*)(D + o_topk),
                (int32_t*)(D + o_out), (int32_t*)(D + o_nm), nn_ratio, check_orientation, m->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(H + o_out, D + o_out, out_bytes, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  for (int p = 0; p < nprob; p++) {
    ps_bf_problem& P = probs[p];
    if (P.nt > 0) memcpy(P.query_of_train, H + o_out + (size_t)dp[p].t_off * 4, (size_t)P.nt * 4);
    P.nmatches = ((const int32_t*)(H + o_nm))[p];
  }
  return PS_OK;
}

}  // extern "C"
