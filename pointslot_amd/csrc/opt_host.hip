// Host side of the optimiser C-ABI (include/pointslot_hip.h): marshals the caller's per-frame / per-object
// problems into one device arena, launches the persistent LM kernels, copies the results back.
// Replaces Optimizer::PoseOptimization / CFSE3ObjStateOptimization / ObjectLocalBundleAdjustment
// (/root/reference/src/Optimizer.cc:249-1075) and the g2o machinery underneath.
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "opt_plan.h"
#include "ps_common.h"
#include "se3.h"
#include "dyn_plan.h"

extern "C" void psk_pose_lm_launch(const PoProb*, int, const PoVertex*, const float*, const float*, const float*,
                                   const uint8_t*, uint8_t*, double*, uint8_t*, void*, double*, int32_t*, double*, hipStream_t);

struct BaCtx { uint8_t* d_buf = nullptr; size_t d_bytes = 0; uint8_t* h_buf = nullptr; size_t h_bytes = 0; };

struct ps_optimizer {
  // One handle may be shared by threads (the shim's process-wide handle serves PoseOptimization on the tracking thread and
  // ObjectLocalBundleAdjustment on the ObjectLocalMapping thread, ObjectLocalMapping.cpp:375-377): calls on it are serialised
  std::mutex mu;
  BaCtx ba;
  int device = 0;
  hipStream_t stream = nullptr;
  uint8_t* d_buf = nullptr; size_t d_bytes = 0;
  uint8_t* h_buf = nullptr; size_t h_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_kernel_ms = 0;
  bool trace = false;
  std::vector<double> last_trace;   // [nprob][PS_PO_TRACE][3]
};

namespace {
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
int ensure(ps_optimizer* m, size_t need) {
  // 25 % headroom: batch sizes drift from call to call, and re-allocating pinned memory costs milliseconds
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

struct PoView {   // one pose-only problem as the packer sees it
  int k;
  const int32_t* off;       // [k+1] (nullptr for k == 1: single range [0, n))
  int n;
  const float *x, *obs, *is2;
  const uint8_t* valid;
  uint8_t* outlier;
  float fx, fy, cx, cy, bf;
  int mode;
};

// shared driver of ps_pose_optimize_batch / ps_cfse3_optimize_batch
int run_pose_only(ps_optimizer* m, const std::vector<PoView>& pv, std::vector<Se3>& poses, std::vector<int32_t>& results) {
  const int nprob = (int)pv.size();
  size_t ne = 0, nv = 0;
  for (const PoView& p : pv) { ne += p.n; nv += p.k; }
  size_t off = 0;
  const size_t o_prob = off; off += al(sizeof(PoProb) * nprob);
  const size_t o_vert = off; off += al(sizeof(PoVertex) * (nv + 1));
  const size_t o_x = off;    off += al(ne * 12 + 16);
  const size_t o_obs = off;  off += al(ne * 12 + 16);
  const size_t o_is2 = off;  off += al(ne * 4 + 16);
  const size_t o_val = off;  off += al(ne + 16);
  const size_t o_out = off;  off += al(ne + 16);
  const size_t o_pose = off; off += al(nv * 56 + 16);
  const size_t in_bytes = off;
  const size_t o_res = off;  off += al((size_t)nprob * 4);
  const size_t o_tr = off;   off += m->trace ? al((size_t)nprob * PS_PO_TRACE * 24) : 0;
  const size_t io_end = off;
  const size_t o_chi = off;  off += al(ne * 8 + 16);
  const size_t o_st = off;   off += al(ne + 16);
  const size_t o_ce = off;   off += al(ne * 32 + 16);     // pose_lm's compacted edge records
  int rc = ensure(m, off);
  if (rc != PS_OK) return rc;
  uint8_t* H = m->h_buf;
  PoProb* hp = (PoProb*)(H + o_prob);
  PoVertex* hv = (PoVertex*)(H + o_vert);
  size_t e0 = 0, v0 = 0;
  for (int p = 0; p < nprob; p++) {
    const PoView& P = pv[p];
    hp[p] = PoProb{(int32_t)v0, P.k, P.mode, P.fx, P.fy, P.cx, P.cy, P.bf};
    for (int o = 0; o < P.k; o++) {
      const int b = P.off ? P.off[o] : 0, e = P.off ? P.off[o + 1] : P.n;
      hv[v0 + o] = PoVertex{(int32_t)(e0 + b), (int32_t)(e0 + e)};
      const Se3& T = poses[v0 + o];
      double* d = (double*)(H + o_pose) + (v0 + o) * 7;
      d[0] = T.t[0]; d[1] = T.t[1]; d[2] = T.t[2]; d[3] = T.q[0]; d[4] = T.q[1]; d[5] = T.q[2]; d[6] = T.q[3];
    }
    if (P.n > 0) {
      memcpy(H + o_x + e0 * 12, P.x, (size_t)P.n * 12);
      memcpy(H + o_obs + e0 * 12, P.obs, (size_t)P.n * 12);
      memcpy(H + o_is2 + e0 * 4, P.is2, (size_t)P.n * 4);
      memcpy(H + o_val + e0, P.valid, (size_t)P.n);
      memcpy(H + o_out + e0, P.outlier, (size_t)P.n);
    }
    e0 += P.n;
    v0 += P.k;
  }
  uint8_t* D = m->d_buf;
  PS_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, m->stream));
  PS_HIP(hipEventRecord(m->ev0, m->stream));
  psk_pose_lm_launch((const PoProb*)(D + o_prob), nprob, (const PoVertex*)(D + o_vert), (const float*)(D + o_x),
                     (const float*)(D + o_obs), (const float*)(D + o_is2), D + o_val, D + o_out, (double*)(D + o_chi),
                     D + o_st, D + o_ce, (double*)(D + o_pose), (int32_t*)(D + o_res), m->trace ? (double*)(D + o_tr) : nullptr,
                     m->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipEventRecord(m->ev1, m->stream));
  PS_HIP(hipMemcpyAsync(H + o_out, D + o_out, io_end - o_out, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  PS_HIP(hipEventElapsedTime(&m->last_kernel_ms, m->ev0, m->ev1));
  e0 = 0;
  for (int p = 0; p < nprob; p++) {
    if (pv[p].n > 0) memcpy(pv[p].outlier, H + o_out + e0, (size_t)pv[p].n);
    e0 += pv[p].n;
  }
  results.assign((const int32_t*)(H + o_res), (const int32_t*)(H + o_res) + nprob);
  for (size_t v = 0; v < nv; v++) {
    const double* d = (const double*)(H + o_pose) + v * 7;
    Se3& T = poses[v];
    T.t[0] = d[0]; T.t[1] = d[1]; T.t[2] = d[2]; T.q[0] = d[3]; T.q[1] = d[4]; T.q[2] = d[5]; T.q[3] = d[6];
  }
  if (m->trace) m->last_trace.assign((const double*)(H + o_tr), (const double*)(H + o_tr) + (size_t)nprob * PS_PO_TRACE * 3);
  return PS_OK;
}
}  // namespace

extern "C" {

int ps_optimizer_create(int device, ps_optimizer** out) {
  if (!out) return ps_set_error(PS_ERR_INVALID, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ps_set_error(PS_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return ps_set_error(PS_ERR_INVALID, "bad device ordinal");
  PS_HIP(hipSetDevice(device));
  ps_optimizer* m = new ps_optimizer();
  m->device = device;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete m; return ps_set_error(PS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
  hipEventCreate(&m->ev0);
  hipEventCreate(&m->ev1);
  *out = m;
  return PS_OK;
}

void ps_optimizer_destroy(ps_optimizer* m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->stream) { hipStreamSynchronize(m->stream); hipStreamDestroy(m->stream); }
  if (m->ev0) hipEventDestroy(m->ev0);
  if (m->ev1) hipEventDestroy(m->ev1);
  if (m->d_buf) hipFree(m->d_buf);
  if (m->h_buf) hipHostFree(m->h_buf);
  if (m->ba.d_buf) hipFree(m->ba.d_buf);
  if (m->ba.h_buf) hipHostFree(m->ba.h_buf);
  delete m;
}

// accessors for ba_host.hip
int psi_optimizer_device(ps_optimizer* m) { return m->device; }
hipStream_t psi_optimizer_stream(ps_optimizer* m) { return m->stream; }
BaCtx* psi_optimizer_ba_ctx(ps_optimizer* m) { return &m->ba; }
void psi_optimizer_set_ms(ps_optimizer* m, float ms) { m->last_kernel_ms = ms; }
std::mutex* psi_optimizer_mutex(ps_optimizer* m) { return &m->mu; }

int ps_optimizer_last_kernel_ms(const ps_optimizer* m, float* ms) {
  if (!m || !ms) return ps_set_error(PS_ERR_INVALID, "null argument");
  *ms = m->last_kernel_ms;
  return PS_OK;
}

int ps_optimizer_enable_trace(ps_optimizer* m, int enable) {
  if (!m) return ps_set_error(PS_ERR_INVALID, "null argument");
  m->trace = enable != 0;
  return PS_OK;
}

int ps_optimizer_get_trace(const ps_optimizer* m, int problem, double* chi2_lambda_trials, int cap, int* n) {
  if (!m || !n || problem < 0 || (size_t)(problem + 1) * PS_PO_TRACE * 3 > m->last_trace.size())
    return ps_set_error(PS_ERR_INVALID, "no trace recorded for problem %d", problem);
  const double* t = &m->last_trace[(size_t)problem * PS_PO_TRACE * 3];
  int cnt = 0;
  while (cnt < PS_PO_TRACE && t[3 * cnt + 2] > 0) cnt++;
  *n = cnt;
  for (int i = 0; i < cnt && i < cap; i++) for (int c = 0; c < 3; c++) chi2_lambda_trials[3 * i + c] = t[3 * i + c];
  return PS_OK;
}

int ps_pose_optimize_batch(ps_optimizer* m, ps_pose_problem* probs, int nprob) {
  if (!m || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_pose_optimize_batch: bad argument");
  std::lock_guard<std::mutex> lock(m->mu);
  PS_HIP(hipSetDevice(m->device));
  std::vector<PoView> pv(nprob);
  std::vector<Se3> poses(nprob);
  for (int p = 0; p < nprob; p++) {
    ps_pose_problem& P = probs[p];
    if (P.n < 0 || (P.n > 0 && (!P.xw || !P.obs || !P.inv_sigma2 || !P.valid || !P.outlier)))
      return ps_set_error(PS_ERR_INVALID, "pose problem %d: null pointers", p);
    pv[p] = PoView{1, nullptr, P.n, P.xw, P.obs, P.inv_sigma2, P.valid, P.outlier, P.fx, P.fy, P.cx, P.cy, P.bf, 0};
    poses[p] = se3_from_mat4f(P.tcw);   // Converter::toSE3Quat(pFrame->mTcw)
  }
  std::vector<int32_t> res;
  int rc = run_pose_only(m, pv, poses, res);
  if (rc != PS_OK) return rc;
  for (int p = 0; p < nprob; p++) {
    probs[p].result = res[p];
    int nvalid = 0;
    for (int i = 0; i < probs[p].n; i++) nvalid += probs[p].valid[i] ? 1 : 0;
    // fewer than 15 correspondences: the reference returns before SetPose (Optimizer.cc:376-377) -> tcw untouched
    if (nvalid >= 15) se3_to_mat4f(poses[p], probs[p].tcw);
  }
  return PS_OK;
}

int ps_cfse3_optimize_batch(ps_optimizer* m, ps_cfse3_problem* probs, int nprob) {
  if (!m || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_cfse3_optimize_batch: bad argument");
  std::lock_guard<std::mutex> lock(m->mu);
  PS_HIP(hipSetDevice(m->device));
  std::vector<PoView> pv;
  std::vector<Se3> poses;
  std::vector<int> map;   // packed problem -> caller problem
  for (int p = 0; p < nprob; p++) {
    ps_cfse3_problem& P = probs[p];
    P.result = 0;
    if (P.k == 0) continue;   // pMObjects.size()==0 -> return false (Optimizer.cc:504-505)
    if (P.k < 0 || P.k > PS_PO_MAX_K || !P.off || !P.poses7)
      return ps_set_error(PS_ERR_INVALID, "cfse3 problem %d: 0 <= k <= %d and non-null off/poses7 required", p, PS_PO_MAX_K);
    const int n = P.off[P.k];
    if (n > 0 && (!P.xo || !P.obs || !P.inv_sigma2 || !P.valid || !P.outlier))
      return ps_set_error(PS_ERR_INVALID, "cfse3 problem %d: null pointers", p);
    pv.push_back(PoView{P.k, P.off, n, P.xo, P.obs, P.inv_sigma2, P.valid, P.outlier, P.fx, P.fy, P.cx, P.cy, P.bf, 1});
    for (int o = 0; o < P.k; o++) {
      Se3 T;
      const double* d = P.poses7 + 7 * o;
      T.t[0] = d[0]; T.t[1] = d[1]; T.t[2] = d[2]; T.q[0] = d[3]; T.q[1] = d[4]; T.q[2] = d[5]; T.q[3] = d[6];
      poses.push_back(T);
    }
    map.push_back(p);
  }
  if (pv.empty()) return PS_OK;
  std::vector<int32_t> res;
  int rc = run_pose_only(m, pv, poses, res);
  if (rc != PS_OK) return rc;
  size_t v = 0;
  for (size_t i = 0; i < pv.size(); i++) {
    ps_cfse3_problem& P = probs[map[i]];
    P.result = res[i];
    for (int o = 0; o < P.k; o++, v++) {
      double* d = P.poses7 + 7 * o;
      const Se3& T = poses[v];
      d[0] = T.t[0]; d[1] = T.t[1]; d[2] = T.t[2]; d[3] = T.q[0]; d[4] = T.q[1]; d[5] = T.q[2]; d[6] = T.q[3];
    }
  }
  return PS_OK;
}

int ps_se3_from_mat4f(const float* m16, double* pose7) {
  if (!m16 || !pose7) return ps_set_error(PS_ERR_INVALID, "null argument");
  const Se3 T = se3_from_mat4f(m16);
  pose7[0] = T.t[0]; pose7[1] = T.t[1]; pose7[2] = T.t[2]; pose7[3] = T.q[0]; pose7[4] = T.q[1]; pose7[5] = T.q[2]; pose7[6] = T.q[3];
  return PS_OK;
}
int ps_se3_to_mat4f(const double* pose7, float* m16) {
  if (!m16 || !pose7) return ps_set_error(PS_ERR_INVALID, "null argument");
  Se3 T;
  T.t[0] = pose7[0]; T.t[1] = pose7[1]; T.t[2] = pose7[2]; T.q[0] = pose7[3]; T.q[1] = pose7[4]; T.q[2] = pose7[5]; T.q[3] = pose7[6];
  se3_to_mat4f(T, m16);
  return PS_OK;
}


extern "C" void psk_dyn_launch(const DynProb*, int, const uint8_t*, const double*, const float*, const float*, double*, int32_t*, hipStream_t);

int ps_dynamic_discrimination_batch(ps_optimizer* h, ps_dyn_problem* probs, int nprob) {
  if (!h || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_dynamic_discrimination_batch: bad argument");
  size_t N = 0;
  for (int p = 0; p < nprob; p++) {
    const ps_dyn_problem& P = probs[p];
    if (P.n < 0 || P.n > PS_DYN_MAX) return ps_set_error(PS_ERR_CAPACITY, "problem %d: 0..%d object points supported", p, PS_DYN_MAX);
    if (P.n > 0 && (!P.valid || !P.po || !P.obs || !P.inv_sigma2)) return ps_set_error(PS_ERR_INVALID, "problem %d: null array", p);
    N += P.n;
  }
  std::lock_guard<std::mutex> lock(h->mu);
  PS_HIP(hipSetDevice(h->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes + 64); return r; };
  const size_t o_prob = take(sizeof(DynProb) * nprob), o_valid = take(N), o_po = take(N * 24), o_obs = take(N * 12), o_is2 = take(N * 4);
  const size_t in_bytes = off;
  const size_t o_avg = take((size_t)nprob * 16), o_n = take((size_t)nprob * 8);
  int rc = ensure(h, off);
  if (rc != PS_OK) return rc;
  uint8_t* H = h->h_buf;
  DynProb* hp = (DynProb*)(H + o_prob);
  size_t n0 = 0;
  auto from7 = [](const double* v) { Se3 T; for (int i = 0; i < 3; i++) T.t[i] = v[i]; for (int i = 0; i < 4; i++) T.q[i] = v[3 + i]; return T; };
  for (int p = 0; p < nprob; p++) {
    const ps_dyn_problem& P = probs[p];
    DynProb& d = hp[p];
    d.off = (int32_t)n0; d.n = P.n;
    d.tco = from7(P.last_tco);
    const Se3 Tl = from7(P.last_tcw), Tc = from7(P.cur_tcw);
    Se3 Tli;   // SE3Quat::inverse (se3quat.h:112-117)
    Tli.q[0] = -Tl.q[0]; Tli.q[1] = -Tl.q[1]; Tli.q[2] = -Tl.q[2]; Tli.q[3] = Tl.q[3];
    const double nt[3] = {Tl.t[0] * -1., Tl.t[1] * -1., Tl.t[2] * -1.};
    se3_rotate(Tli.q, nt, Tli.t);
    d.trel = se3_mul(Tc, Tli);
    d.fx = P.fx; d.fy = P.fy; d.cx = P.cx; d.cy = P.cy; d.mbf = P.mbf;
    if (P.n > 0) {
      memcpy(H + o_valid + n0, P.valid, P.n); memcpy(H + o_po + n0 * 24, P.po, (size_t)P.n * 24);
      memcpy(H + o_obs + n0 * 12, P.obs, (size_t)P.n * 12); memcpy(H + o_is2 + n0 * 4, P.inv_sigma2, (size_t)P.n * 4);
    }
    n0 += P.n;
  }
  uint8_t* D = h->d_buf;
  PS_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, h->stream));
  psk_dyn_launch((const DynProb*)(D + o_prob), nprob, D + o_valid, (const double*)(D + o_po), (const float*)(D + o_obs),
                 (const float*)(D + o_is2), (double*)(D + o_avg), (int32_t*)(D + o_n), h->stream);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(H + o_avg, D + o_avg, off - o_avg, hipMemcpyDeviceToHost, h->stream));
  PS_HIP(hipStreamSynchronize(h->stream));
  for (int p = 0; p < nprob; p++) {
    probs[p].mono_avg = ((const double*)(H + o_avg))[2 * p]; probs[p].stereo_avg = ((const double*)(H + o_avg))[2 * p + 1];
    probs[p].mono_n = ((const int32_t*)(H + o_n))[2 * p]; probs[p].stereo_n = ((const int32_t*)(H + o_n))[2 * p + 1];
  }
  return PS_OK;
}

}  // extern "C"
