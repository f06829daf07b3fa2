// Host side of the matcher C-ABI (include/pointslot_hip.h): packs the caller's per-object problems into
// one upload, launches the kernels of match_kernels.hip on the handle's stream, copies the results back.
// Replaces ORBmatcher::SearchByBruceMatching / DescriptorDistance — /root/reference/src/ORBmatcher.cc.
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <string.h>
#include <vector>
#include "match_plan.h"
#include "ps_common.h"

extern "C" {
void psk_bf_launch(const BfBlock*, int, const BfProb*, int, const uint8_t*, const float*, const uint8_t*, const uint8_t*,
                   const float*, uint32_t*, int32_t*, int32_t*, float, int, hipStream_t);
void psk_hamming_matrix_launch(const uint8_t*, int, const uint8_t*, int, uint16_t*, hipStream_t);
void psk_pj_launch(const PjArrays*, int, int, int, int, int, hipStream_t);
void psk_fuse_launch(const FuArrays*, int, int, hipStream_t);
void psk_distinctive_launch(const uint8_t*, const int32_t*, int32_t*, int, hipStream_t);
}

struct ps_matcher {
  std::mutex mu;              // calls on one handle are serialised (a handle may be shared by threads)
  int device = 0;
  hipStream_t stream = nullptr;
  uint8_t* d_buf = nullptr;   // one growable device arena
  size_t d_bytes = 0;
  uint8_t* h_buf = nullptr;   // pinned staging of the same size
  size_t h_bytes = 0;
  uint8_t* d_wide = nullptr;  // second arena, only for the wide re-run of search problems whose windows overflowed the candidate store
  size_t d_wide_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the kernels of the last brute-force / projection call (ps_matcher_last_kernel_ms)
  float last_kernel_ms = 0;
};

namespace {
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

int ensure(ps_matcher* m, size_t need) {
  // 25 % headroom: batch sizes drift from call to call (keypoint counts), and re-allocating pinned memory costs milliseconds
  const size_t bytes = (need > m->d_bytes || need > m->h_bytes) ? al(need + need / 4) : need;
  if (bytes > m->d_bytes) {
    if (m->d_buf) hipFree(m->d_buf);
    m->d_buf = nullptr;
    PS_HIP(hipMalloc(&m->d_buf, bytes));
    if (const char* fill = getenv("PS_DEBUG_FILL")) {   // diagnostic: poison fresh device memory.  hipMemset on device memory returns before the fill has run, and it runs
      PS_HIP(hipMemset(m->d_buf, atoi(fill), bytes));                 // on the null stream, which the handle's non-blocking stream does not wait for: without the wait the fill
      PS_HIP(hipDeviceSynchronize());        // landed on top of the call's uploads now and then (r06: 2 of 50 runs of the poisoned test slice died of it)
    }
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
  hipEventCreate(&m->ev0); hipEventCreate(&m->ev1);
  *out = m;
  return PS_OK;
}

int ps_matcher_last_kernel_ms(const ps_matcher* m, float* ms) {
  if (!m || !ms) return ps_set_error(PS_ERR_INVALID, "null argument");
  *ms = m->last_kernel_ms;
  return PS_OK;
}

void ps_matcher_destroy(ps_matcher* m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->ev0) hipEventDestroy(m->ev0);
  if (m->ev1) hipEventDestroy(m->ev1);
  if (m->stream) { hipStreamSynchronize(m->stream); hipStreamDestroy(m->stream); }
  if (m->d_buf) hipFree(m->d_buf);
  if (m->d_wide) hipFree(m->d_wide);
  if (m->h_buf) hipHostFree(m->h_buf);
  delete m;
}

int ps_hamming_matrix(ps_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
  if (!m || !q || !t || !out || nq < 1 || nt < 1) return ps_set_error(PS_ERR_INVALID, "ps_hamming_matrix: bad argument");
  std::lock_guard<std::mutex> lock(m->mu);
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
  std::lock_guard<std::mutex> lock(m->mu);
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
  hipEventRecord(m->ev0, m->stream);
  psk_bf_launch((const BfBlock*)(D + o_blk), (int)blocks.size(), (const BfProb*)(D + o_prob), nprob, D + o_qd,
                (const float*)(D + o_qa), D + o_qv, D + o_td, (const float*)(D + o_ta), (uint32_t*)(D + o_topk),
                (int32_t*)(D + o_out), (int32_t*)(D + o_nm), nn_ratio, check_orientation, m->stream);
  hipEventRecord(m->ev1, m->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(H + o_out, D + o_out, out_bytes, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  hipEventElapsedTime(&m->last_kernel_ms, m->ev0, m->ev1);
  for (int p = 0; p < nprob; p++) {
    ps_bf_problem& P = probs[p];
    if (P.nt > 0) memcpy(P.query_of_train, H + o_out + (size_t)dp[p].t_off * 4, (size_t)P.nt * 4);
    P.nmatches = ((const int32_t*)(H + o_nm))[p];
  }
  return PS_OK;
}


// The three ORBmatcher::SearchByProjection overloads (see include/pointslot_hip.h for the field mapping).
int ps_search_by_projection(ps_matcher* m, ps_proj_problem* probs, int nprob) {
  if (!m || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_search_by_projection: bad argument");
  std::lock_guard<std::mutex> lock(m->mu);
  PS_HIP(hipSetDevice(m->device));
  size_t NT = 0, NQ = 0;
  int max_nq = 0, max_nt = 0, any_frame = 0;
  const int NCELL = PS_GRID_COLS * PS_GRID_ROWS;
  for (int p = 0; p < nprob; p++) {
    const ps_proj_problem& P = probs[p];
    const ps_proj_train& T = P.train;
    if (T.n < 0 || T.n > 32767 || P.nq < 0 || (T.n > 0 && (!T.x || !T.y || !T.octave || !T.angle || !T.u_right || !T.desc ||
        !T.occupied || !T.cell_off || !T.cell_idx || !P.match_of_train)))
      return ps_set_error(PS_ERR_INVALID, "projection problem %d: bad train side (n <= 32767)", p);
    if (P.nq > 0 && (!P.q_valid || !P.q_desc || !P.q_observed)) return ps_set_error(PS_ERR_INVALID, "projection problem %d: null query arrays", p);
    if (P.frame_mode) {
      if (P.nq > 0 && (!P.q_xw || !P.q_octave || !P.q_angle)) return ps_set_error(PS_ERR_INVALID, "projection problem %d: frame mode needs q_xw/q_octave/q_angle", p);
      any_frame = 1;
    } else if (P.nq > 0 && (!P.q_u || !P.q_v || !P.q_ur || !P.q_radius || !P.q_radius_er || !P.q_min_level || !P.q_max_level))
      return ps_set_error(PS_ERR_INVALID, "projection problem %d: null pre-projected query arrays", p);
    if (P.use_bbox && T.n > 0 && !T.in_bbox) return ps_set_error(PS_ERR_INVALID, "projection problem %d: use_bbox without in_bbox", p);
    NT += T.n; NQ += P.nq;
    max_nq = P.nq > max_nq ? P.nq : max_nq;
    max_nt = T.n > max_nt ? T.n : max_nt;
  }
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes + 64); return r; };
  const size_t o_prob = take(sizeof(PjProb) * nprob);
  const size_t o_tx = take(NT * 4), o_ty = take(NT * 4), o_toct = take(NT * 4), o_tang = take(NT * 4), o_tur = take(NT * 4);
  const size_t o_tdesc = take(NT * 32), o_tocc = take(NT), o_tbb = take(NT), o_coff = take((size_t)nprob * (NCELL + 1) * 4), o_cidx = take(NT * 4);
  const size_t o_qvalid = take(NQ), o_qu = take(NQ * 4), o_qv = take(NQ * 4), o_qur = take(NQ * 4), o_qrad = take(NQ * 4), o_qrer = take(NQ * 4);
  const size_t o_qminl = take(NQ * 4), o_qmaxl = take(NQ * 4), o_qdesc = take(NQ * 32), o_qobs = take(NQ), o_qang = take(NQ * 4);
  const size_t o_qxw = take(NQ * 12), o_qoct = take(NQ * 4);
  const size_t in_bytes = off;
  const size_t o_match = take(NT * 4), o_nm = take((size_t)nprob * 4), o_ovf = take((size_t)nprob * 4);
  const size_t out_end = off;
  const size_t o_cand = take(NQ * PS_PJ_CAP * 4), o_ncand = take(NQ * 4), o_qbest = take(NQ * 4), o_qbin = take(NQ), o_tt = take(NQ * 16);
  int rc = ensure(m, off);
  if (rc != PS_OK) return rc;
  uint8_t* H = m->h_buf;
  static const bool prof = getenv("PS_MATCH_PROFILE") != nullptr;   // developer switch: host-side breakdown of the call on stderr
  const auto tp0 = std::chrono::steady_clock::now();
  PjProb* hp = (PjProb*)(H + o_prob);
  std::vector<size_t> toff(nprob), qoff(nprob);
  {
    size_t t0 = 0, q0 = 0;
    for (int p = 0; p < nprob; p++) { toff[p] = t0; qoff[p] = q0; t0 += probs[p].train.n; q0 += probs[p].nq; }
  }
  // every problem fills (or zeroes) exactly its own slices of the staging buffer: independent, memcpy-bound
  ps_parallel_for(nprob, in_bytes, [&](int p) {
    const ps_proj_problem& P = probs[p];
    const ps_proj_train& T = P.train;
    const size_t t0 = toff[p], q0 = qoff[p];
    PjProb& d = hp[p];
    memset(&d, 0, sizeof(d));
    d.t_off = (int32_t)t0; d.nt = T.n; d.q_off = (int32_t)q0; d.c_off = (int32_t)q0; d.nq = P.nq; d.grid_off = p * (NCELL + 1);
    d.min_x = T.min_x; d.min_y = T.min_y; d.gw_inv = T.grid_w_inv; d.gh_inv = T.grid_h_inv;
    d.th_dist = P.th_dist; d.ratio_test = P.ratio_test; d.nn_ratio = P.nn_ratio; d.check_ori = P.check_orientation;
    d.use_bbox = P.use_bbox; d.frame_mode = P.frame_mode;
    memcpy(d.tcw, P.tcw, 64); memcpy(d.tlw, P.tlw, 64);
    d.fx = P.fx; d.fy = P.fy; d.cx = P.cx; d.cy = P.cy; d.mbf = P.mbf; d.mb = P.mb;
    memcpy(d.bounds, P.bounds, 16); memcpy(d.scale, P.scale_factors, 32);
    d.th = P.th; d.mono = P.mono;
    if (T.n > 0) {
      const size_t n = (size_t)T.n, filled = (size_t)T.cell_off[NCELL];
      memcpy(H + o_tx + t0 * 4, T.x, n * 4); memcpy(H + o_ty + t0 * 4, T.y, n * 4);
      memcpy(H + o_toct + t0 * 4, T.octave, n * 4); memcpy(H + o_tang + t0 * 4, T.angle, n * 4);
      memcpy(H + o_tur + t0 * 4, T.u_right, n * 4); memcpy(H + o_tdesc + t0 * 32, T.desc, n * 32);
      memcpy(H + o_tocc + t0, T.occupied, n);
      if (T.in_bbox) memcpy(H + o_tbb + t0, T.in_bbox, n); else memset(H + o_tbb + t0, 0, n);
      memcpy(H + o_coff + (size_t)d.grid_off * 4, T.cell_off, (size_t)(NCELL + 1) * 4);
      memcpy(H + o_cidx + t0 * 4, T.cell_idx, std::min(filled, n) * 4);
      if (filled < n) memset(H + o_cidx + (t0 + filled) * 4, 0, (n - filled) * 4);
    } else {
      memset(H + o_coff + (size_t)d.grid_off * 4, 0, (size_t)(NCELL + 1) * 4);
    }
    if (P.nq > 0) {
      const size_t n = (size_t)P.nq;
      memcpy(H + o_qvalid + q0, P.q_valid, n); memcpy(H + o_qdesc + q0 * 32, P.q_desc, n * 32);
      memcpy(H + o_qobs + q0, P.q_observed, n);
      if (P.q_angle) memcpy(H + o_qang + q0 * 4, P.q_angle, n * 4); else memset(H + o_qang + q0 * 4, 0, n * 4);
      const size_t pre[7] = {o_qu, o_qv, o_qur, o_qrad, o_qrer, o_qminl, o_qmaxl};
      if (P.frame_mode) {
        memcpy(H + o_qxw + q0 * 12, P.q_xw, n * 12); memcpy(H + o_qoct + q0 * 4, P.q_octave, n * 4);
        for (size_t o : pre) memset(H + o + q0 * 4, 0, n * 4);
      } else {
        const void* src[7] = {P.q_u, P.q_v, P.q_ur, P.q_radius, P.q_radius_er, P.q_min_level, P.q_max_level};
        for (int k = 0; k < 7; k++) memcpy(H + pre[k] + q0 * 4, src[k], n * 4);
        memset(H + o_qxw + q0 * 12, 0, n * 12); memset(H + o_qoct + q0 * 4, 0, n * 4);
      }
    }
  });
  uint8_t* D = m->d_buf;
  const auto tp1 = std::chrono::steady_clock::now();
  PS_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, m->stream));
  if (prof) PS_HIP(hipStreamSynchronize(m->stream));
  const auto tp2 = std::chrono::steady_clock::now();
  PS_HIP(hipMemsetAsync(D + o_ovf, 0, (size_t)nprob * 4, m->stream));
  PjArrays A;
  A.prob = (const PjProb*)(D + o_prob);
  A.tx = (const float*)(D + o_tx); A.ty = (const float*)(D + o_ty); A.toct = (const int32_t*)(D + o_toct);
  A.tang = (const float*)(D + o_tang); A.tur = (const float*)(D + o_tur); A.tdesc = D + o_tdesc; A.tocc = D + o_tocc;
  A.tbbox = D + o_tbb; A.cell_off = (const int32_t*)(D + o_coff); A.cell_idx = (const int32_t*)(D + o_cidx);
  A.qvalid = D + o_qvalid; A.qu = (float*)(D + o_qu); A.qv = (float*)(D + o_qv); A.qur = (float*)(D + o_qur);
  A.qrad = (float*)(D + o_qrad); A.qrer = (float*)(D + o_qrer); A.qminl = (int32_t*)(D + o_qminl); A.qmaxl = (int32_t*)(D + o_qmaxl);
  A.qdesc = D + o_qdesc; A.qobs = D + o_qobs; A.qang = (const float*)(D + o_qang);
  A.qxw = (const float*)(D + o_qxw); A.qoct = (const int32_t*)(D + o_qoct);
  A.cand = (uint32_t*)(D + o_cand); A.ncand = (int32_t*)(D + o_ncand); A.match = (int32_t*)(D + o_match);
  A.nmatch = (int32_t*)(D + o_nm); A.overflow = (int32_t*)(D + o_ovf); A.qbest = (int32_t*)(D + o_qbest); A.qbin = D + o_qbin;
  A.ttop = (uint4*)(D + o_tt);
  hipEventRecord(m->ev0, m->stream);
  psk_pj_launch(&A, nprob, max_nq > 0 ? max_nq : 1, max_nt, any_frame, 0, m->stream);
  hipEventRecord(m->ev1, m->stream);
  PS_HIP(hipGetLastError());
  if (prof) PS_HIP(hipStreamSynchronize(m->stream));
  const auto tp3 = std::chrono::steady_clock::now();
  PS_HIP(hipMemcpyAsync(H + o_match, D + o_match, out_end - o_match, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  hipEventElapsedTime(&m->last_kernel_ms, m->ev0, m->ev1);
  const auto tp4 = std::chrono::steady_clock::now();
  if (prof) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "ps_search_by_projection %d problems, %.1f MB in: pack %.3f ms, upload %.3f, kernels %.3f, read-back %.3f\n", nprob, in_bytes / 1e6, ms(tp0, tp1),
            ms(tp1, tp2), ms(tp2, tp3), ms(tp3, tp4));
  }
  // Problems with a window of more than PS_PJ_CAP candidates (dense imagery, the 2 * th retry at a coarse octave) run again with
  // the wide key format - 1024 candidates per window, frames of up to 8191 features - in a compact candidate store of their own;
  // the other problems of the batch keep their results.  A problem that overflows that too (or has more features) is the only one
  // that fails: nmatches = -1, its match_of_train untouched, and the call returns PS_ERR_CAPACITY after serving the rest.
  const int32_t* ovf = (const int32_t*)(H + o_ovf);
  std::vector<int> wide, failed;
  for (int p = 0; p < nprob; p++)
    if (ovf[p] > 0) (probs[p].train.n <= PS_PJ_WIDE_MAX_N ? wide : failed).push_back(p);
  std::vector<int32_t> wide_nm(wide.size(), 0);
  if (!wide.empty()) {
    size_t nq2 = 0;
    int max_nq2 = 1, any_frame2 = 0;
    std::vector<PjProb> sub(wide.size());
    for (size_t i = 0; i < wide.size(); i++) {
      sub[i] = hp[wide[i]];
      sub[i].c_off = (int32_t)nq2;
      nq2 += probs[wide[i]].nq;
      max_nq2 = std::max(max_nq2, probs[wide[i]].nq);
      any_frame2 |= probs[wide[i]].frame_mode;
    }
    size_t woff = 0;
    auto wtake = [&](size_t bytes) { size_t r = woff; woff += al(bytes + 64); return r; };
    const size_t w_prob = wtake(sizeof(PjProb) * sub.size()), w_nm = wtake(sub.size() * 4), w_ovf = wtake(sub.size() * 4), w_cand = wtake(nq2 * PS_PJ_CAP_WIDE * 4);
    if (woff > m->d_wide_bytes) {
      if (m->d_wide) hipFree(m->d_wide);
      m->d_wide = nullptr; m->d_wide_bytes = 0;
      PS_HIP(hipMalloc(&m->d_wide, woff));
      m->d_wide_bytes = woff;
    }
    uint8_t* Wd = m->d_wide;
    PS_HIP(hipMemcpyAsync(Wd + w_prob, sub.data(), sizeof(PjProb) * sub.size(), hipMemcpyHostToDevice, m->stream));
    PS_HIP(hipMemsetAsync(Wd + w_ovf, 0, sub.size() * 4, m->stream));
    // (pj_project runs again on these problems: it only repeats what it wrote; the queries it invalidated stay invalid)
    PjArrays A2 = A;
    A2.prob = (const PjProb*)(Wd + w_prob); A2.cand = (uint32_t*)(Wd + w_cand); A2.nmatch = (int32_t*)(Wd + w_nm); A2.overflow = (int32_t*)(Wd + w_ovf);
    psk_pj_launch(&A2, (int)sub.size(), max_nq2, max_nt, any_frame2, 1, m->stream);
    PS_HIP(hipGetLastError());
    std::vector<int32_t> ovf2(sub.size(), 0);
    PS_HIP(hipMemcpyAsync(H + o_match, D + o_match, NT * 4, hipMemcpyDeviceToHost, m->stream));
    PS_HIP(hipStreamSynchronize(m->stream));
    PS_HIP(hipMemcpy(wide_nm.data(), Wd + w_nm, sub.size() * 4, hipMemcpyDeviceToHost));
    PS_HIP(hipMemcpy(ovf2.data(), Wd + w_ovf, sub.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < wide.size(); i++)
      if (ovf2[i] > 0) { failed.push_back(wide[i]); wide_nm[i] = -1; }
  }
  for (int p = 0; p < nprob; p++) {
    ps_proj_problem& P = probs[p];
    if (std::find(failed.begin(), failed.end(), p) != failed.end()) { P.nmatches = -1; continue; }
    if (P.train.n > 0) memcpy(P.match_of_train, H + o_match + toff[p] * 4, (size_t)P.train.n * 4);
    P.nmatches = ((const int32_t*)(H + o_nm))[p];
  }
  for (size_t i = 0; i < wide.size(); i++)
    if (wide_nm[i] >= 0) probs[wide[i]].nmatches = wide_nm[i];
  if (!failed.empty())
    return ps_set_error(PS_ERR_CAPACITY, "projection problem %d (and %zu more): a search window held more than %d candidates%s", failed[0], failed.size() - 1,
                        probs[failed[0]].train.n <= PS_PJ_WIDE_MAX_N ? PS_PJ_CAP_WIDE : PS_PJ_CAP,
                        probs[failed[0]].train.n <= PS_PJ_WIDE_MAX_N ? "" : " and the frame has more features than the wide candidate store indexes (8191)");
  return PS_OK;
}


int ps_distinctive_descriptors(ps_matcher* m, const uint8_t* desc, const int32_t* off, int npoints, int32_t* best) {
  if (!m || !off || !best || npoints < 1) return ps_set_error(PS_ERR_INVALID, "ps_distinctive_descriptors: bad argument");
  const int total = off[npoints];
  if (total > 0 && !desc) return ps_set_error(PS_ERR_INVALID, "null descriptors");
  for (int p = 0; p < npoints; p++)
    if (off[p + 1] < off[p] || off[p + 1] - off[p] > 128) return ps_set_error(PS_ERR_CAPACITY, "point %d: 0..128 observations supported", p);
  std::lock_guard<std::mutex> lock(m->mu);
  PS_HIP(hipSetDevice(m->device));
  const size_t o_desc = 0, o_off = al((size_t)total * 32 + 32), o_best = o_off + al((size_t)(npoints + 1) * 4), end = o_best + al((size_t)npoints * 4);
  int rc = ensure(m, end);
  if (rc != PS_OK) return rc;
  if (total > 0) memcpy(m->h_buf + o_desc, desc, (size_t)total * 32);
  memcpy(m->h_buf + o_off, off, (size_t)(npoints + 1) * 4);
  PS_HIP(hipMemcpyAsync(m->d_buf, m->h_buf, o_best, hipMemcpyHostToDevice, m->stream));
  psk_distinctive_launch(m->d_buf + o_desc, (const int32_t*)(m->d_buf + o_off), (int32_t*)(m->d_buf + o_best), npoints, m->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(m->h_buf + o_best, m->d_buf + o_best, (size_t)npoints * 4, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  memcpy(best, m->h_buf + o_best, (size_t)npoints * 4);
  return PS_OK;
}


int ps_fuse_search(ps_matcher* m, ps_fuse_problem* probs, int nprob) {
  if (!m || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_fuse_search: bad argument");
  std::lock_guard<std::mutex> lock(m->mu);
  PS_HIP(hipSetDevice(m->device));
  size_t NT = 0, NQ = 0;
  int max_nq = 0;
  const int NCELL = PS_GRID_COLS * PS_GRID_ROWS;
  for (int p = 0; p < nprob; p++) {
    const ps_fuse_problem& P = probs[p];
    const ps_proj_train& T = P.train;
    if (T.n < 0 || P.nq < 0 || (T.n > 0 && (!T.x || !T.y || !T.octave || !T.u_right || !T.desc || !T.cell_off || !T.cell_idx)))
      return ps_set_error(PS_ERR_INVALID, "fuse problem %d: bad keyframe side", p);
    if (P.nq > 0 && (!P.q_valid || !P.q_pos || !P.q_normal || !P.q_min_dist || !P.q_max_dist || !P.q_desc || !P.best_idx || !P.best_dist))
      return ps_set_error(PS_ERR_INVALID, "fuse problem %d: null query arrays", p);
    if (P.n_levels < 1 || P.n_levels > 8) return ps_set_error(PS_ERR_INVALID, "fuse problem %d: n_levels must be 1..8", p);
    NT += T.n; NQ += P.nq;
    max_nq = P.nq > max_nq ? P.nq : max_nq;
  }
  if (NQ == 0) return PS_OK;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes + 64); return r; };
  const size_t o_prob = take(sizeof(FuProb) * nprob);
  const size_t o_tx = take(NT * 4), o_ty = take(NT * 4), o_toct = take(NT * 4), o_tur = take(NT * 4), o_tdesc = take(NT * 32);
  const size_t o_coff = take((size_t)nprob * (NCELL + 1) * 4), o_cidx = take(NT * 4);
  const size_t o_qvalid = take(NQ), o_qpos = take(NQ * 12), o_qnor = take(NQ * 12), o_qmin = take(NQ * 4), o_qmax = take(NQ * 4), o_qdesc = take(NQ * 32);
  const size_t in_bytes = off;
  const size_t o_bi = take(NQ * 4), o_bd = take(NQ * 4);
  int rc = ensure(m, off);
  if (rc != PS_OK) return rc;
  uint8_t* H = m->h_buf;
  memset(H, 0, in_bytes);
  FuProb* hp = (FuProb*)(H + o_prob);
  size_t t0 = 0, q0 = 0;
  for (int p = 0; p < nprob; p++) {
    const ps_fuse_problem& P = probs[p];
    const ps_proj_train& T = P.train;
    FuProb& d = hp[p];
    d.t_off = (int32_t)t0; d.nt = T.n; d.q_off = (int32_t)q0; d.nq = P.nq; d.grid_off = p * (NCELL + 1);
    d.min_x = T.min_x; d.min_y = T.min_y; d.gw_inv = T.grid_w_inv; d.gh_inv = T.grid_h_inv;
    memcpy(d.R, P.rcw, 36); memcpy(d.t, P.tcw, 12); memcpy(d.ow, P.ow, 12);
    d.fx = P.fx; d.fy = P.fy; d.cx = P.cx; d.cy = P.cy; d.bf = P.bf;
    memcpy(d.bounds, P.bounds, 32); memcpy(d.scale, P.scale_factors, 32); memcpy(d.inv_sigma2, P.inv_level_sigma2, 32);
    d.log_scale = P.log_scale_factor; d.n_levels = P.n_levels; d.th = P.th;
    if (T.n > 0) {
      memcpy(H + o_tx + t0 * 4, T.x, (size_t)T.n * 4); memcpy(H + o_ty + t0 * 4, T.y, (size_t)T.n * 4);
      memcpy(H + o_toct + t0 * 4, T.octave, (size_t)T.n * 4); memcpy(H + o_tur + t0 * 4, T.u_right, (size_t)T.n * 4);
      memcpy(H + o_tdesc + t0 * 32, T.desc, (size_t)T.n * 32);
      memcpy(H + o_cidx + t0 * 4, T.cell_idx, (size_t)T.cell_off[NCELL] * 4);
    }
    if (T.cell_off) memcpy(H + o_coff + (size_t)d.grid_off * 4, T.cell_off, (size_t)(NCELL + 1) * 4);
    if (P.nq > 0) {
      memcpy(H + o_qvalid + q0, P.q_valid, P.nq); memcpy(H + o_qpos + q0 * 12, P.q_pos, (size_t)P.nq * 12);
      memcpy(H + o_qnor + q0 * 12, P.q_normal, (size_t)P.nq * 12); memcpy(H + o_qmin + q0 * 4, P.q_min_dist, (size_t)P.nq * 4);
      memcpy(H + o_qmax + q0 * 4, P.q_max_dist, (size_t)P.nq * 4); memcpy(H + o_qdesc + q0 * 32, P.q_desc, (size_t)P.nq * 32);
    }
    t0 += T.n; q0 += P.nq;
  }
  uint8_t* D = m->d_buf;
  PS_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, m->stream));
  FuArrays A;
  A.prob = (const FuProb*)(D + o_prob);
  A.tx = (const float*)(D + o_tx); A.ty = (const float*)(D + o_ty); A.toct = (const int32_t*)(D + o_toct); A.tur = (const float*)(D + o_tur);
  A.tdesc = D + o_tdesc; A.cell_off = (const int32_t*)(D + o_coff); A.cell_idx = (const int32_t*)(D + o_cidx);
  A.qvalid = D + o_qvalid; A.qpos = (const float*)(D + o_qpos); A.qnormal = (const float*)(D + o_qnor); A.qmin = (const float*)(D + o_qmin);
  A.qmax = (const float*)(D + o_qmax); A.qdesc = D + o_qdesc; A.best_idx = (int32_t*)(D + o_bi); A.best_dist = (int32_t*)(D + o_bd);
  psk_fuse_launch(&A, nprob, max_nq, m->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(H + o_bi, D + o_bi, off - o_bi, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  q0 = 0;
  for (int p = 0; p < nprob; p++) {
    ps_fuse_problem& P = probs[p];
    if (P.nq > 0) {
      memcpy(P.best_idx, H + o_bi + q0 * 4, (size_t)P.nq * 4);
      memcpy(P.best_dist, H + o_bd + q0 * 4, (size_t)P.nq * 4);
    }
    q0 += P.nq;
  }
  return PS_OK;
}

}  // extern "C"
