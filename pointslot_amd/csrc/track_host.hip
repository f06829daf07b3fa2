// Host side of the device-resident lockstep tracker (include/pointslot_hip.h: ps_tracker_*): S independent stereo sequences
// advance one frame per call; the whole per-frame chain of the tracking thread (/root/reference/src/Tracking.cc:2840-3160:
// Frame::Frame, TrackWithMotionModel, TrackLocalMap) is queued on one stream — extractor, stereo matcher, the windowed
// matchers, the pose optimiser and the glue kernels of track_kernels.hip in between — and nothing comes back to the host
// until ps_tracker_fetch.  Same results as pointslot_amd/host/StereoOdometry.h driving the per-call C-ABI.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "orb_plan.h"
#include "ps_common.h"
#include "track_plan.h"

extern "C" {
hipStream_t psi_orb_stream(ps_orb*);
const OrbPlan* psi_orb_plan(ps_orb*);
int psi_orb_prepare(ps_orb*, int, int);
void psk_pj_launch(const PjArrays*, int, int, int, int, hipStream_t);
void psk_pose_lm_launch(const PoProb*, int, const PoVertex*, const float*, const float*, const float*, const uint8_t*, uint8_t*,
                        double*, uint8_t*, double*, int32_t*, double*, hipStream_t);
void psk_trk_begin(const TrkArrays*, int, hipStream_t);
void psk_trk_after_mm1(const TrkArrays*, hipStream_t);
void psk_trk_after_mm(const TrkArrays*, int, hipStream_t);
void psk_trk_after_pose1(const TrkArrays*, int, hipStream_t);
void psk_trk_after_lm(const TrkArrays*, hipStream_t);
void psk_trk_finish(const TrkArrays*, int, hipStream_t);
}

namespace {
enum { TS_ORB = 0, TS_STEREO, TS_GLUE, TS_SEARCH, TS_POSE, TS_COUNT };
const char* kTrkStage[TS_COUNT] = {"orb_extract", "stereo_match", "track_glue", "search_by_projection", "pose_optimization"};
// events of one step: boundaries between the launches below, and the stage every interval belongs to
const int kTrkIntervals = 12;
const int kTrkIntervalStage[kTrkIntervals] = {TS_ORB, TS_STEREO, TS_GLUE, TS_SEARCH, TS_GLUE, TS_SEARCH, TS_GLUE, TS_POSE, TS_GLUE, TS_SEARCH, TS_GLUE, TS_POSE};
}  // namespace

struct ps_tracker {
  ps_tracker_config cfg;
  ps_orb* orb = nullptr;
  hipStream_t stream = nullptr;
  uint8_t* d_buf = nullptr;      // one arena for everything below
  size_t d_bytes = 0;
  TrkArrays A;
  PjArrays pj_mm1, pj_mm2, pj_lm;
  // pose optimiser work arrays
  double* po_chi2 = nullptr; uint8_t* po_state = nullptr;
  int32_t* d_overflow = nullptr;
  int step = 0;
  float mb = 0, mbf = 0;
  // stage timing
  static const int RING = 64;
  hipEvent_t ev[RING][kTrkIntervals + 2] = {};
  bool timing = false;
  int timed = 0;
};

namespace {
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

struct Carver {
  uint8_t* base; size_t off = 0;
  template <typename T> T* take(size_t count) { T* p = base ? (T*)(base + off) : nullptr; off += al(count * sizeof(T) + 64); return p; }
};

void carve_frame(Carver& c, TrkFrame& f, size_t S, size_t cap) {
  const size_t n = S * cap;
  f.x = c.take<float>(n); f.y = c.take<float>(n); f.angle = c.take<float>(n); f.uright = c.take<float>(n); f.depth = c.take<float>(n);
  f.xw = c.take<float>(3 * n); f.octave = c.take<int32_t>(n); f.mp_id = c.take<int32_t>(n);
  f.cell_off = c.take<int32_t>(S * (PS_TRK_NCELL + 1)); f.cell_idx = c.take<int32_t>(n);
  f.desc = c.take<uint8_t>(32 * n); f.mp_valid = c.take<uint8_t>(n); f.mp_observed = c.take<uint8_t>(n); f.outlier = c.take<uint8_t>(n);
  f.n = c.take<int32_t>(S); f.tcw = c.take<float>(16 * S);
}

// lays the tracker's arrays out in `base` (nullptr: only measures); returns the bytes needed
size_t carve(ps_tracker* t, uint8_t* base) {
  Carver c{base};
  TrkArrays& A = t->A;
  const size_t S = A.S, cap = A.cap, n = S * cap;
  carve_frame(c, A.cur, S, cap); carve_frame(c, A.last, S, cap);
  A.seq = c.take<TrkSeq>(S);
  A.lm_xw = c.take<float>(3 * n); A.lm_normal = c.take<float>(3 * n); A.lm_maxd = c.take<float>(n); A.lm_mind = c.take<float>(n); A.lm_desc = c.take<uint8_t>(32 * n);
  A.prob_mm1 = c.take<PjProb>(S); A.prob_mm2 = c.take<PjProb>(S); A.prob_lm = c.take<PjProb>(S);
  A.nmatch_mm1 = c.take<int32_t>(S); A.nmatch_mm2 = c.take<int32_t>(S); A.nmatch_lm = c.take<int32_t>(S);
  A.qvalid = c.take<uint8_t>(n); A.occupied = c.take<uint8_t>(n); A.match = c.take<int32_t>(n);
  A.qu = c.take<float>(n); A.qv = c.take<float>(n); A.qur = c.take<float>(n); A.qrad = c.take<float>(n);
  A.qminl = c.take<int32_t>(n); A.qmaxl = c.take<int32_t>(n);
  A.po_prob = c.take<PoProb>(S); A.po_vert = c.take<PoVertex>(S); A.po_obs = c.take<float>(3 * n); A.po_is2 = c.take<float>(n);
  A.po_pose = c.take<double>(7 * S); A.po_result = c.take<int32_t>(S);
  A.traj = c.take<float>((size_t)A.max_steps * S * 16);
  A.stats = c.take<TrkStat>((size_t)A.max_steps * S);
  t->po_chi2 = c.take<double>(n); t->po_state = c.take<uint8_t>(n);
  t->d_overflow = c.take<int32_t>(S);   // per sequence, accumulated over the searches of all steps
  // windowed-matcher work arrays (one set: the three searches of a step run one after the other)
  uint8_t* ones = c.take<uint8_t>(n);
  uint32_t* cand = c.take<uint32_t>(n * PS_PJ_CAP);
  int32_t* ncand = c.take<int32_t>(n);
  int32_t* qbest = c.take<int32_t>(n);
  uint8_t* qbin = c.take<uint8_t>(n);
  uint4* ttop = c.take<uint4>(n);
  PjArrays P;
  memset(&P, 0, sizeof(P));
  P.tx = A.cur.x; P.ty = A.cur.y; P.toct = A.cur.octave; P.tang = A.cur.angle; P.tur = A.cur.uright; P.tdesc = A.cur.desc;
  P.tocc = A.occupied; P.tbbox = A.occupied; P.cell_off = A.cur.cell_off; P.cell_idx = A.cur.cell_idx;
  P.qvalid = A.qvalid; P.qu = A.qu; P.qv = A.qv; P.qur = A.qur; P.qrad = A.qrad; P.qrer = A.qrad; P.qminl = A.qminl; P.qmaxl = A.qmaxl;
  P.qobs = ones; P.qang = A.last.angle; P.qxw = A.last.xw; P.qoct = A.last.octave;
  P.cand = cand; P.ncand = ncand; P.match = A.match; P.overflow = t->d_overflow; P.qbest = qbest; P.ttop = ttop; P.qbin = qbin;
  t->pj_mm1 = P; t->pj_mm1.prob = A.prob_mm1; t->pj_mm1.qdesc = A.last.desc; t->pj_mm1.nmatch = A.nmatch_mm1;
  t->pj_mm2 = P; t->pj_mm2.prob = A.prob_mm2; t->pj_mm2.qdesc = A.last.desc; t->pj_mm2.nmatch = A.nmatch_mm2;
  t->pj_lm = P;  t->pj_lm.prob = A.prob_lm;   t->pj_lm.qdesc = A.lm_desc;    t->pj_lm.nmatch = A.nmatch_lm;
  if (base) {
    // q_observed is 1 for every query of both searches (OdoSequence: qobs.assign(n, 1))
    hipMemsetAsync(ones, 1, n, t->stream);
  }
  return c.off;
}

int queue_chain(ps_tracker* t) {
  const TrkArrays* A = &t->A;
  hipStream_t st = t->stream;
  const int S = A->S, cap = A->cap;
  const bool tm = t->timing;
  hipEvent_t* ev = t->ev[t->timed % ps_tracker::RING];
  int e = 1;   // ev[0] was recorded before the extraction, ev[1] after it comes first here
  auto mark = [&]() { if (tm) hipEventRecord(ev[e++], st); };
  mark();                                                       // end of orb
  int rc = ps_orb_stereo_match_batch(t->orb, S, t->mb, t->mbf);
  if (rc != PS_OK) return rc;
  mark();                                                       // stereo
  psk_trk_begin(A, t->step, st); mark();                        // glue
  psk_pj_launch(&t->pj_mm1, S, cap, 1, 0, st); mark();             // search
  psk_trk_after_mm1(A, st); mark();                             // glue
  psk_pj_launch(&t->pj_mm2, S, cap, 1, 0, st); mark();             // search (the 2 * th retry; empty problems where it is not needed)
  psk_trk_after_mm(A, t->step, st); mark();                     // glue
  psk_pose_lm_launch(A->po_prob, S, A->po_vert, A->cur.xw, A->po_obs, A->po_is2, A->cur.mp_valid, A->cur.outlier, t->po_chi2, t->po_state,
                     A->po_pose, A->po_result, nullptr, st);
  mark();                                                       // pose
  psk_trk_after_pose1(A, t->step, st); mark();                  // glue
  psk_pj_launch(&t->pj_lm, S, cap, 0, 0, st); mark();              // search
  psk_trk_after_lm(A, st); mark();                              // glue
  psk_pose_lm_launch(A->po_prob, S, A->po_vert, A->cur.xw, A->po_obs, A->po_is2, A->cur.mp_valid, A->cur.outlier, t->po_chi2, t->po_state,
                     A->po_pose, A->po_result, nullptr, st);
  mark();                                                       // pose
  psk_trk_finish(A, t->step, st);
  if (tm) { hipEventRecord(ev[e++], st); t->timed++; }          // glue (the last interval is folded into the previous glue slot below)
  PS_HIP(hipGetLastError());
  t->step++;
  return PS_OK;
}
}  // namespace

extern "C" {

int ps_tracker_create(const ps_tracker_config* cfg, ps_tracker** out) {
  if (!cfg || !out) return ps_set_error(PS_ERR_INVALID, "ps_tracker_create: null argument");
  if (cfg->n_sequences < 1 || cfg->width < 1 || cfg->height < 1 || cfg->max_steps < 1 || !(cfg->fx > 0) || !(cfg->fy > 0) || !(cfg->bf > 0) || !(cfg->th_depth > 0))
    return ps_set_error(PS_ERR_INVALID, "ps_tracker_create: bad configuration");
  ps_orb_config oc{cfg->nfeatures, cfg->scale_factor, cfg->nlevels, cfg->ini_th_fast, cfg->min_th_fast, 2 * cfg->n_sequences, cfg->device};
  ps_orb* orb = nullptr;
  int rc = ps_orb_create(&oc, &orb);
  if (rc != PS_OK) return rc;
  rc = psi_orb_prepare(orb, cfg->width, cfg->height);
  if (rc != PS_OK) { ps_orb_destroy(orb); return rc; }
  const OrbPlan* plan = psi_orb_plan(orb);
  if (plan->kp_cap > 4096) { ps_orb_destroy(orb); return ps_set_error(PS_ERR_CAPACITY, "the tracker supports at most 4096 keypoints per image (stereo matcher)"); }
  ps_tracker* t = new ps_tracker();
  t->cfg = *cfg;
  t->orb = orb;
  t->stream = psi_orb_stream(orb);   // the extractor's stream: its kernels and the chain behind them are ordered without events
  TrkArrays& A = t->A;
  memset(&A, 0, sizeof(A));
  A.S = cfg->n_sequences; A.cap = plan->kp_cap; A.kp_cap = plan->kp_cap; A.max_steps = cfg->max_steps;
  TrkCam& C = A.cam;
  std::vector<float> sf(8, 1.f), is2(8, 1.f);
  ps_orb_get_tables(orb, sf.data(), nullptr, nullptr, is2.data(), nullptr);
  C.fx = cfg->fx; C.fy = cfg->fy; C.cx = cfg->cx; C.cy = cfg->cy; C.mbf = cfg->bf;
  C.mb = C.mbf / C.fx;                                             // Frame.cc: mb = mbf / fx
  C.th_depth = C.mbf * cfg->th_depth / C.fx;                       // Tracking.cc:402
  C.w = cfg->width; C.h = cfg->height; C.nlevels = cfg->nlevels;
  C.gw_inv = (float)PS_GRID_COLS / (float)cfg->width; C.gh_inv = (float)PS_GRID_ROWS / (float)cfg->height;   // Frame.cc:1636-1640
  for (int l = 0; l < 8; l++) { C.sf[l] = l < cfg->nlevels ? sf[l] : 1.f; C.inv_sigma2[l] = l < cfg->nlevels ? is2[l] : 1.f; }
  C.log_sf = logf(C.sf[cfg->nlevels > 1 ? 1 : 0]);
  C.inv_fx = 1.f / C.fx; C.inv_fy = 1.f / C.fy;
  t->mb = C.mb; t->mbf = C.mbf;
  const size_t bytes = carve(t, nullptr);
  hipError_t e = hipMalloc(&t->d_buf, bytes);
  if (e != hipSuccess) { ps_orb_destroy(orb); delete t; return ps_set_error(PS_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
  t->d_bytes = bytes;
  hipMemsetAsync(t->d_buf, 0, bytes, t->stream);   // every sequence NOT_INITIALIZED, phase idle
  carve(t, t->d_buf);
  const float *d_ur = nullptr, *d_dp = nullptr; const int32_t* d_kept = nullptr;
  const ps_keypoint* d_kps = nullptr; const uint8_t* d_desc = nullptr; const int32_t* d_cnt = nullptr; int32_t kc = 0;
  ps_orb_batch_device_outputs(orb, &d_kps, &d_desc, &d_cnt, &kc);
  ps_orb_stereo_device_outputs(orb, &d_ur, &d_dp, &d_kept);
  A.orb_kps = d_kps; A.orb_desc = d_desc; A.orb_counts = d_cnt; A.orb_uright = d_ur; A.orb_depth = d_dp;
  for (int r = 0; r < ps_tracker::RING; r++)
    for (int i = 0; i < kTrkIntervals + 2; i++) hipEventCreate(&t->ev[r][i]);
  if (hipStreamSynchronize(t->stream) != hipSuccess) { ps_tracker_destroy(t); return ps_set_error(PS_ERR_HIP, "tracker initialisation failed"); }
  *out = t;
  return PS_OK;
}

void ps_tracker_destroy(ps_tracker* t) {
  if (!t) return;
  hipSetDevice(t->cfg.device);
  if (t->stream) hipStreamSynchronize(t->stream);
  for (int r = 0; r < ps_tracker::RING; r++)
    for (int i = 0; i < kTrkIntervals + 2; i++) if (t->ev[r][i]) hipEventDestroy(t->ev[r][i]);
  if (t->d_buf) hipFree(t->d_buf);
  ps_orb_destroy(t->orb);
  delete t;
}

int ps_tracker_orb(ps_tracker* t, ps_orb** orb) {
  if (!t || !orb) return ps_set_error(PS_ERR_INVALID, "null argument");
  *orb = t->orb;
  return PS_OK;
}

int ps_tracker_step_device(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch) {
  if (!t || !d_imgs) return ps_set_error(PS_ERR_INVALID, "ps_tracker_step_device: null argument");
  if (t->step >= t->A.max_steps) return ps_set_error(PS_ERR_CAPACITY, "the tracker was created for %d steps", t->A.max_steps);
  PS_HIP(hipSetDevice(t->cfg.device));
  if (t->timing) PS_HIP(hipEventRecord(t->ev[t->timed % ps_tracker::RING][0], t->stream));
  int rc = ps_orb_extract_batch_device(t->orb, d_imgs, 2 * t->A.S, t->cfg.width, t->cfg.height, stride, image_pitch, nullptr);
  if (rc != PS_OK) return rc;
  return queue_chain(t);
}

int ps_tracker_step(ps_tracker* t, const uint8_t* const* left, const uint8_t* const* right, int stride) {
  if (!t || !left || !right) return ps_set_error(PS_ERR_INVALID, "ps_tracker_step: null argument");
  if (t->step >= t->A.max_steps) return ps_set_error(PS_ERR_CAPACITY, "the tracker was created for %d steps", t->A.max_steps);
  PS_HIP(hipSetDevice(t->cfg.device));
  std::vector<const uint8_t*> imgs(2 * (size_t)t->A.S);
  for (int k = 0; k < t->A.S; k++) { imgs[2 * k] = left[k]; imgs[2 * k + 1] = right[k]; }
  if (t->timing) PS_HIP(hipEventRecord(t->ev[t->timed % ps_tracker::RING][0], t->stream));
  int rc = ps_orb_extract_batch(t->orb, imgs.data(), 2 * t->A.S, t->cfg.width, t->cfg.height, stride);
  if (rc != PS_OK) return rc;
  return queue_chain(t);
}

int ps_tracker_sync(ps_tracker* t) {
  if (!t) return ps_set_error(PS_ERR_INVALID, "null handle");
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  return PS_OK;
}

int ps_tracker_steps(const ps_tracker* t, int* steps) {
  if (!t || !steps) return ps_set_error(PS_ERR_INVALID, "null argument");
  *steps = t->step;
  return PS_OK;
}

int ps_tracker_fetch(ps_tracker* t, int first_step, int nsteps, float* tcw, ps_track_stat* stats) {
  if (!t || first_step < 0 || nsteps < 0 || first_step + nsteps > t->step) return ps_set_error(PS_ERR_INVALID, "ps_tracker_fetch: steps [%d, %d) of %d", first_step, first_step + nsteps, t ? t->step : 0);
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  static_assert(sizeof(ps_track_stat) == sizeof(TrkStat), "ps_track_stat layout");
  const size_t S = t->A.S;
  if (tcw && nsteps) PS_HIP(hipMemcpy(tcw, t->A.traj + (size_t)first_step * S * 16, (size_t)nsteps * S * 64, hipMemcpyDeviceToHost));
  if (stats && nsteps) PS_HIP(hipMemcpy(stats, t->A.stats + (size_t)first_step * S, (size_t)nsteps * S * sizeof(TrkStat), hipMemcpyDeviceToHost));
  std::vector<int32_t> ovf(S, 0);
  PS_HIP(hipMemcpy(ovf.data(), t->d_overflow, S * 4, hipMemcpyDeviceToHost));
  long total = 0;
  int first = -1;
  for (size_t k = 0; k < S; k++) { total += ovf[k]; if (ovf[k] && first < 0) first = (int)k; }
  if (total > 0)
    return ps_set_error(PS_ERR_CAPACITY, "%ld search window(s) held more than %d candidates (first in sequence %d): use the per-call matcher, which re-runs such "
                        "problems with the wide candidate store", total, PS_PJ_CAP, first);
  return PS_OK;
}

int ps_tracker_reset(ps_tracker* t) {
  if (!t) return ps_set_error(PS_ERR_INVALID, "null handle");
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  PS_HIP(hipMemsetAsync(t->A.seq, 0, sizeof(TrkSeq) * t->A.S, t->stream));
  PS_HIP(hipMemsetAsync(t->d_overflow, 0, sizeof(int32_t) * t->A.S, t->stream));
  t->step = 0;
  return PS_OK;
}

int ps_tracker_enable_stage_timing(ps_tracker* t, int enable) {
  if (!t) return ps_set_error(PS_ERR_INVALID, "null handle");
  t->timing = enable != 0;
  t->timed = 0;
  return PS_OK;
}

int ps_tracker_stage_times(ps_tracker* t, const char** names, float* ms, int cap, int* n) {
  if (!t || !n) return ps_set_error(PS_ERR_INVALID, "null argument");
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  const int cnt = t->timed < ps_tracker::RING ? t->timed : ps_tracker::RING;
  double acc[TS_COUNT] = {};
  for (int r = 0; r < cnt; r++) {
    for (int i = 0; i < kTrkIntervals; i++) {
      float v = 0;
      PS_HIP(hipEventElapsedTime(&v, t->ev[r][i], t->ev[r][i + 1]));
      acc[kTrkIntervalStage[i]] += v;
    }
    float v = 0;   // trk_finish
    PS_HIP(hipEventElapsedTime(&v, t->ev[r][kTrkIntervals], t->ev[r][kTrkIntervals + 1]));
    acc[TS_GLUE] += v;
  }
  *n = TS_COUNT;
  for (int i = 0; i < TS_COUNT && i < cap; i++) {
    if (names) names[i] = kTrkStage[i];
    if (ms) ms[i] = cnt ? (float)(acc[i] / cnt) : 0.f;
  }
  return PS_OK;
}

}  // extern "C"
