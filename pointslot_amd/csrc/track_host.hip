// Host side of the device-resident lockstep tracker (include/pointslot_hip.h: ps_tracker_*): S independent stereo sequences
// advance one frame per call; the whole per-frame chain of the tracking thread (/root/reference/src/Tracking.cc:2840-3160:
// Frame::Frame, TrackWithMotionModel, TrackLocalMap) is queued on one stream — extractor, stereo matcher, the windowed
// matchers, the pose optimiser and the glue kernels of track_kernels.hip in between — and nothing comes back to the host
// until ps_tracker_fetch.  Same results as pointslot_amd/host/StereoOdometry.h driving the per-call C-ABI.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "objtrack_plan.h"
#include "orb_plan.h"
#include "ps_common.h"
#include "track_plan.h"

extern "C" {
hipStream_t psi_orb_stream(ps_orb*);
const OrbPlan* psi_orb_plan(ps_orb*);
int psi_orb_prepare(ps_orb*, int, int);
void psk_pj_launch(const PjArrays*, int, int, int, int, int, hipStream_t);
void psk_pose_lm_launch(const PoProb*, int, const PoVertex*, const float*, const float*, const float*, const uint8_t*, uint8_t*,
                        double*, uint8_t*, void*, double*, int32_t*, double*, hipStream_t);
void psk_trk_begin(const TrkArrays*, int, hipStream_t);
void psk_trk_after_mm1(const TrkArrays*, hipStream_t);
void psk_trk_after_mm(const TrkArrays*, int, hipStream_t);
void psk_trk_after_pose1(const TrkArrays*, int, hipStream_t);
void psk_trk_after_lm(const TrkArrays*, hipStream_t);
void psk_trk_finish(const TrkArrays*, int, hipStream_t);
void psk_trk_stamp_overflow(const TrkArrays*, int32_t*, int, hipStream_t);
uint8_t* psi_orb_arena(ps_orb*);
void psk_stereo_launch(const OrbPlan*, const StPair*, int, int, float, float, hipStream_t);
void psk_bf_launch(const BfBlock*, int, const BfProb*, int, const uint8_t*, const float*, const uint8_t*, const uint8_t*, const float*, uint32_t*, int32_t*, int32_t*,
                   float, int, hipStream_t);
void psk_ob_masks(const ObArrays*, uint8_t*, int, int, int, uint8_t*, int, int, hipStream_t);
int psi_cvorb_batch_begin(ps_cvorb*, int, int, int, hipStream_t, uint8_t**, int*, int*);
int psi_cvorb_batch_run(ps_cvorb*, const uint8_t*, const uint8_t*, int, int, size_t, int, size_t, int, hipStream_t);
void psk_ob_begin(const ObArrays*, int, hipStream_t);
void psk_ob_track(const ObArrays*, int, hipStream_t);
void psk_ob_bf_blocks(const BfProb*, BfBlock*, int32_t*, int, int, hipStream_t);
void psk_bf_launch_dev(const BfBlock*, const int32_t*, int, const BfProb*, int, const uint8_t*, const float*, const uint8_t*, const uint8_t*, const float*, uint32_t*,
                       int32_t*, int32_t*, float, int, hipStream_t);
void psk_ob_after_bf(const ObArrays*, int, hipStream_t);
void psk_ob_after_cf1(const ObArrays*, int, hipStream_t);
void psk_ob_after_lm(const ObArrays*, int, hipStream_t);
void psk_ob_finish(const ObArrays*, int, hipStream_t);
}

namespace {
enum { TS_ORB = 0, TS_STEREO, TS_GLUE, TS_SEARCH, TS_POSE, TS_OBJ_FEATURES, TS_OBJ_STEREO, TS_OBJ_GLUE, TS_OBJ_BRUTEFORCE, TS_OBJ_CFSE3, TS_OBJ_SEARCH, TS_OBJ_WAIT, TS_COUNT };
const char* kTrkStage[TS_COUNT] = {"orb_extract", "stereo_match", "track_glue", "search_by_projection", "pose_optimization",
                                   "object_features", "object_stereo_match", "object_glue", "object_bruteforce", "object_cfse3", "object_search_by_projection",
                                   "object_features_wait"};
const int kTrkMaxEvents = 32;   // events of one step: boundaries between the launches of queue_chain
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
  double* po_chi2 = nullptr; uint8_t* po_state = nullptr; float* po_cedge = nullptr;   // cedge: pose_lm's compacted edge records, 32 B per slot
  int32_t* d_overflow = nullptr;
  int step = 0;
  float mb = 0, mbf = 0;
  // the object half (max_objects > 0)
  ps_cvorb* cvorb = nullptr;
  uint8_t* d_obj = nullptr; size_t d_obj_bytes = 0;
  ObArrays OA;
  uint8_t* d_objmask = nullptr;      // [2 S][h][w] LeftObjMask / RightObjMask
  StPair* d_obj_pairs = nullptr;
  BfBlock* d_bf_blocks = nullptr; int32_t* d_bf_count = nullptr; int bf_blocks_per_prob = 0;
  PjArrays pj_obj;
  double* ob_chi2 = nullptr; uint8_t* ob_state = nullptr; float* ob_cedge = nullptr;
  // ExtractObjORB does not depend on the camera chain of its frame: with the overlap on, the head of the object chain - masks, cv::ORB,
  // ComputeObjStereoMatches (behind the extractor's pyramids), ob_begin - runs on a second (low-priority) stream beside the camera chain
  // and joins before TrackMapObject.  Measured with 512 sequences (r03): 16.16 against 16.34 ms per step - the kernels of both streams
  // slow each other down by what the overlap saves - so a throughput-sized handle keeps one stream, where the stage times add up to the
  // step.  A handle of a few sequences (r06: BASELINE configs[4] is ONE sequence per GPU) leaves the GPU almost empty, every kernel runs
  // at its own latency, and the overlap takes the head's ~0.3 ms off the frame: default on up to 32 sequences (PS_TRK_OVERLAP=0 / 1 decides).
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_orb = nullptr;
  hipEvent_t ev_obj[64][2] = {};     // timing of the object features on stream2 (ring as below)
  bool overlap = false;
  // stage timing
  static const int RING = 64;
  hipEvent_t ev[RING][kTrkMaxEvents] = {};
  int ev_stage[kTrkMaxEvents] = {};   // stage of the interval that ends at event i (the same for every step)
  int ev_count = 0;
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
  t->po_chi2 = c.take<double>(n); t->po_state = c.take<uint8_t>(n); t->po_cedge = c.take<float>(8 * n);
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

// carve the object half's arrays (nullptr base: only measures)
void carve_obj_frame(Carver& c, ObFrame& f, size_t S, size_t OC, size_t K) {
  const size_t n = S * OC;
  f.x = c.take<float>(n); f.y = c.take<float>(n); f.angle = c.take<float>(n); f.uright = c.take<float>(n); f.depth = c.take<float>(n);
  f.octave = c.take<int32_t>(n); f.desc = c.take<uint8_t>(32 * n);
  f.mp_valid = c.take<uint8_t>(n); f.mp_observed = c.take<uint8_t>(n); f.outlier = c.take<uint8_t>(n); f.mp_id = c.take<int32_t>(n); f.mp_po = c.take<float>(3 * n);
  f.off = c.take<int32_t>(S * (K + 1)); f.det = c.take<ObDet>(S * K); f.mo = c.take<int32_t>(S * K); f.tco = c.take<double>(S * K * 7);
  f.cell_off = c.take<int32_t>(S * K * (OB_NCELL + 1)); f.cell_idx = c.take<int32_t>(n); f.ndet = c.take<int32_t>(S);
}

size_t carve_obj(ps_tracker* t, uint8_t* base) {
  Carver c{base};
  ObArrays& A = t->OA;
  const size_t S = A.S, K = A.K, M = A.M, OC = A.OC, LC = A.LC, n = S * OC, nl = S * M * LC;
  carve_obj_frame(c, A.cur, S, OC, K); carve_obj_frame(c, A.last, S, OC, K);
  A.mobj = c.take<ObMapObject>(S * M);
  A.lm_po = c.take<float>(3 * nl); A.lm_normal = c.take<float>(3 * nl); A.lm_maxd = c.take<float>(nl); A.lm_mind = c.take<float>(nl); A.lm_desc = c.take<uint8_t>(32 * nl);
  A.owner = c.take<int8_t>(n);
  A.in_last = c.take<int32_t>(S * K); A.tracked = c.take<int32_t>(S * K); A.need = c.take<int32_t>(S * K); A.track_ok = c.take<int32_t>(S * K);
  A.inl_flag = c.take<int32_t>(n); A.cam_pts = c.take<double>(3 * n); A.last_tco = c.take<double>(S * K * 7);
  A.bf_prob = c.take<BfProb>(S * K); A.bf_topk = c.take<uint32_t>(n * PS_BF_TOPK); A.bf_qvalid = c.take<uint8_t>(n); A.bf_qot = c.take<int32_t>(n);
  A.bf_nmatch = c.take<int32_t>(S * K);
  A.pj_prob = c.take<PjProb>(S * K);
  A.pj_qvalid = c.take<uint8_t>(nl); A.pj_qu = c.take<float>(nl); A.pj_qv = c.take<float>(nl); A.pj_qur = c.take<float>(nl); A.pj_qrad = c.take<float>(nl);
  A.pj_qrer = c.take<float>(nl); A.pj_qminl = c.take<int32_t>(nl); A.pj_qmaxl = c.take<int32_t>(nl);
  A.occupied = c.take<uint8_t>(n); A.inbbox = c.take<uint8_t>(n); A.pj_match = c.take<int32_t>(n); A.pj_nmatch = c.take<int32_t>(S * K);
  A.po_prob = c.take<PoProb>(S); A.po_vert = c.take<PoVertex>(S * K + 1); A.po_obs = c.take<float>(3 * n); A.po_is2 = c.take<float>(n);
  A.po_pose = c.take<double>(S * K * 7); A.po_result = c.take<int32_t>(S); A.po_vmap = c.take<int32_t>(S * K);
  A.stats = c.take<ObStat>((size_t)A.max_steps * S * K);
  A.dropped = c.take<int32_t>(S);
  A.det_overflow = c.take<int32_t>(S); A.search_overflow = c.take<int32_t>(S);
  t->ob_chi2 = c.take<double>(n); t->ob_state = c.take<uint8_t>(n); t->ob_cedge = c.take<float>(8 * n);
  t->d_objmask = c.take<uint8_t>(2 * S * (size_t)((t->cfg.width + 15) & ~15) * t->cfg.height);
  t->d_obj_pairs = c.take<StPair>(S);
  t->bf_blocks_per_prob = (int)((OC + PS_BF_QPB - 1) / PS_BF_QPB);
  t->d_bf_blocks = c.take<BfBlock>(S * K * t->bf_blocks_per_prob);
  t->d_bf_count = c.take<int32_t>(16);
  float* st_ur = c.take<float>(n); float* st_dp = c.take<float>(n); int32_t* st_sad = c.take<int32_t>(n); int32_t* st_kept = c.take<int32_t>(S);
  uint8_t* st_scratch = c.take<uint8_t>(S * (size_t)PS_ST_SCRATCH);
  // windowed-matcher work arrays of the object searches
  uint8_t* ones = c.take<uint8_t>(nl);
  uint32_t* cand = c.take<uint32_t>(S * K * LC * PS_PJ_CAP);
  int32_t* ncand = c.take<int32_t>(nl); int32_t* qbest = c.take<int32_t>(nl); uint8_t* qbin = c.take<uint8_t>(nl); uint4* ttop = c.take<uint4>(nl);
  int32_t* ovf = c.take<int32_t>(S * K);
  A.pj_overflow = ovf;
  A.st_uright = st_ur; A.st_depth = st_dp;
  PjArrays P;
  memset(&P, 0, sizeof(P));
  P.prob = A.pj_prob;
  P.tx = A.cur.x; P.ty = A.cur.y; P.toct = A.cur.octave; P.tang = A.cur.angle; P.tur = A.cur.uright; P.tdesc = A.cur.desc;
  P.tocc = A.occupied; P.tbbox = A.inbbox; P.cell_off = A.cur.cell_off; P.cell_idx = A.cur.cell_idx;
  P.qvalid = A.pj_qvalid; P.qu = A.pj_qu; P.qv = A.pj_qv; P.qur = A.pj_qur; P.qrad = A.pj_qrad; P.qrer = A.pj_qrer; P.qminl = A.pj_qminl; P.qmaxl = A.pj_qmaxl;
  P.qdesc = A.lm_desc; P.qobs = ones; P.qang = A.pj_qu; P.qxw = nullptr; P.qoct = nullptr;
  P.cand = cand; P.ncand = ncand; P.match = A.pj_match; P.nmatch = A.pj_nmatch; P.overflow = ovf; P.qbest = qbest; P.ttop = ttop; P.qbin = qbin;
  t->pj_obj = P;
  if (base) {
    hipMemsetAsync(ones, 1, nl, t->stream);
    // the stereo matcher's pair table of the object keys: cv::ORB results of images 2s / 2s + 1 against the extractor's pyramids
    const OrbPlan* plan = psi_orb_plan(t->orb);
    uint8_t* arena = psi_orb_arena(t->orb);
    const ps_keypoint* ckps = nullptr; const uint8_t* cdesc = nullptr; const int32_t* ccnt = nullptr; int32_t ccap = 0;
    const int32_t* covf = nullptr;
    ps_cvorb_batch_device_outputs(t->cvorb, &ckps, &cdesc, &ccnt, &covf, &ccap);
    A.cv_overflow = covf;
    std::vector<StPair> pairs(S);
    for (size_t k = 0; k < S; k++) {
      StPair& p = pairs[k];
      const size_t l = 2 * k, r = l + 1;
      p.arena_l = arena + l * plan->arena_bytes; p.arena_r = arena + r * plan->arena_bytes;
      p.kps_l = ckps + l * ccap; p.desc_l = cdesc + l * ccap * 32; p.cnt_l = ccnt + l;
      p.kps_r = ckps + r * ccap; p.desc_r = cdesc + r * ccap * 32; p.cnt_r = ccnt + r;
      p.u_right = st_ur + k * OC; p.depth = st_dp + k * OC; p.sad = st_sad + k * OC; p.kept = st_kept + k;
      p.scratch = st_scratch + k * PS_ST_SCRATCH;
    }
    hipMemcpy(t->d_obj_pairs, pairs.data(), S * sizeof(StPair), hipMemcpyHostToDevice);
    A.cv_kps = ckps; A.cv_desc = cdesc; A.cv_count = ccnt; A.cv_cap = ccap;
  }
  return c.off;
}

// ExtractObjORB (Frame.cc:711, 2623-2665) with its masks (Frame.cc:2318-2503) on stream `so`
int queue_object_features(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch, const uint8_t* d_masks, int mask_stride, size_t mask_pitch,
                          const ps_detection* d_dets, hipStream_t so) {
  ObArrays* O = &t->OA;
  const int S = t->A.S;
  O->idmask = d_masks; O->mask_stride = mask_stride; O->mask_pitch = mask_pitch; O->dets_in = (const ObDet*)d_dets;
  const int W = t->cfg.width, H = t->cfg.height;
  uint8_t* occ = nullptr; int ocw = 0, och = 0;
  int r = psi_cvorb_batch_begin(t->cvorb, 2 * S, W, H, so, &occ, &ocw, &och);
  if (r != PS_OK) return r;
  const int ostride = (W + 15) & ~15;                          // rows of the object masks start on 16-byte boundaries
  psk_ob_masks(O, t->d_objmask, W, H, ostride, occ, ocw, och, so);   // ... and fills the detector's cell occupancy on the way
  return psi_cvorb_batch_run(t->cvorb, d_imgs, t->d_objmask, 2 * S, stride, image_pitch, ostride, (size_t)ostride * H, 1, so);
}

// The head of the object chain on the second stream, in two parts around the extraction the caller queues on the main stream: the object
// features start with the frame (they read the images and the masks only); ComputeObjStereoMatches reads the extractor's pyramids and
// waits for them; ob_begin follows.  The main stream joins in front of TrackMapObject (queue_chain).
int overlap_head_before_orb(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch, const uint8_t* d_masks, int mask_stride, size_t mask_pitch,
                            const ps_detection* d_dets) {
  PS_HIP(hipEventRecord(t->ev_fork, t->stream));                // behind the previous frame's chain, which still reads this stream's arrays
  PS_HIP(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
  if (t->timing) hipEventRecord(t->ev_obj[t->timed % ps_tracker::RING][0], t->stream2);
  return queue_object_features(t, d_imgs, stride, image_pitch, d_masks, mask_stride, mask_pitch, d_dets, t->stream2);
}
int overlap_head_after_orb(ps_tracker* t) {
  PS_HIP(hipEventRecord(t->ev_orb, t->stream));
  PS_HIP(hipStreamWaitEvent(t->stream2, t->ev_orb, 0));
  psk_stereo_launch(psi_orb_plan(t->orb), t->d_obj_pairs, t->A.S, t->OA.OC, t->mb, t->mbf, t->stream2);
  psk_ob_begin(&t->OA, t->step, t->stream2);
  if (t->timing) hipEventRecord(t->ev_obj[t->timed % ps_tracker::RING][1], t->stream2);
  PS_HIP(hipEventRecord(t->ev_join, t->stream2));
  return PS_OK;
}

int queue_chain(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch, const uint8_t* d_masks, int mask_stride, size_t mask_pitch,
                const ps_detection* d_dets) {
  TrkArrays* A = &t->A;
  hipStream_t st = t->stream;
  const int S = A->S, cap = A->cap;
  const bool tm = t->timing;
  hipEvent_t* ev = t->ev[t->timed % ps_tracker::RING];
  int e = 1;   // ev[0] was recorded before the extraction
  static const bool dbg = getenv("PS_TRK_DEBUG_SYNC") != nullptr;   // diagnostic: wait after every launch group and say which one it was
  auto mark = [&](int stage) {
    t->ev_stage[e] = stage;
    if (tm) hipEventRecord(ev[e], st);
    if (dbg) { const hipError_t r = hipStreamSynchronize(st); fprintf(stderr, "[ps_tracker] step %d: launch group %d (%s) done: %s\n", t->step, e, kTrkStage[stage], hipGetErrorString(r)); }
    e++;
  };
  A->idmask = d_masks; A->mask_stride = mask_stride; A->mask_pitch = mask_pitch;
  mark(TS_ORB);
  int rc = PS_OK;
  const bool objects = d_masks && t->cvorb;
  auto object_features = [&](hipStream_t so) -> int { return queue_object_features(t, d_imgs, stride, image_pitch, d_masks, mask_stride, mask_pitch, d_dets, so); };
  rc = ps_orb_stereo_match_batch(t->orb, S, t->mb, t->mbf);
  if (rc != PS_OK) return rc;
  mark(TS_STEREO);
  psk_trk_begin(A, t->step, st); mark(TS_GLUE);
  psk_pj_launch(&t->pj_mm1, S, cap, cap, 1, 0, st); mark(TS_SEARCH);
  psk_trk_after_mm1(A, st); mark(TS_GLUE);
  psk_pj_launch(&t->pj_mm2, S, cap, cap, 1, 0, st); mark(TS_SEARCH);   // the 2 * th retry; empty problems where it is not needed
  psk_trk_after_mm(A, t->step, st); mark(TS_GLUE);
  psk_pose_lm_launch(A->po_prob, S, A->po_vert, A->cur.xw, A->po_obs, A->po_is2, A->cur.mp_valid, A->cur.outlier, t->po_chi2, t->po_state,
                     t->po_cedge, A->po_pose, A->po_result, nullptr, st);
  mark(TS_POSE);
  psk_trk_after_pose1(A, t->step, st); mark(TS_GLUE);
  psk_pj_launch(&t->pj_lm, S, cap, cap, 0, 0, st); mark(TS_SEARCH);
  psk_trk_after_lm(A, st); mark(TS_GLUE);
  psk_pose_lm_launch(A->po_prob, S, A->po_vert, A->cur.xw, A->po_obs, A->po_is2, A->cur.mp_valid, A->cur.outlier, t->po_chi2, t->po_state,
                     t->po_cedge, A->po_pose, A->po_result, nullptr, st);
  mark(TS_POSE);
  psk_trk_finish(A, t->step, st);
  psk_trk_stamp_overflow(A, t->d_overflow, t->step, st); mark(TS_GLUE);
  if (objects) {
    // ---- the object half of Tracking::Track, behind the camera chain of the same frame ----
    ObArrays* O = &t->OA;
    const int K = O->K;
    if (t->overlap) { PS_HIP(hipStreamWaitEvent(st, t->ev_join, 0)); mark(TS_OBJ_WAIT); }   // the head ran on the second stream (overlap_head_*)
    else {
      rc = object_features(st);
      if (rc != PS_OK) return rc;
      mark(TS_OBJ_FEATURES);
      psk_stereo_launch(psi_orb_plan(t->orb), t->d_obj_pairs, S, O->OC, t->mb, t->mbf, st); mark(TS_OBJ_STEREO);
      psk_ob_begin(O, t->step, st);
    }
    psk_ob_track(O, t->step, st);
    psk_ob_bf_blocks(O->bf_prob, t->d_bf_blocks, t->d_bf_count, S * K, t->bf_blocks_per_prob, st); mark(TS_OBJ_GLUE);
    psk_bf_launch_dev(t->d_bf_blocks, t->d_bf_count, 1024, O->bf_prob, S * K, O->last.desc, O->last.angle, O->bf_qvalid, O->cur.desc, O->cur.angle,
                      O->bf_topk, O->bf_qot, O->bf_nmatch, 0.9f, 1, st);                   // ORBmatcher matcher(0.9, true), Tracking.cc:2381
    mark(TS_OBJ_BRUTEFORCE);
    psk_ob_after_bf(O, t->step, st); mark(TS_OBJ_GLUE);
    psk_pose_lm_launch(O->po_prob, S, O->po_vert, O->cur.mp_po, O->po_obs, O->po_is2, O->cur.mp_valid, O->cur.outlier, t->ob_chi2, t->ob_state,
                       t->ob_cedge, O->po_pose, O->po_result, nullptr, st);
    mark(TS_OBJ_CFSE3);
    psk_ob_after_cf1(O, t->step, st); mark(TS_OBJ_GLUE);
    psk_pj_launch(&t->pj_obj, S * K, O->LC, O->OC, 0, 0, st); mark(TS_OBJ_SEARCH);
    psk_ob_after_lm(O, t->step, st); mark(TS_OBJ_GLUE);
    psk_pose_lm_launch(O->po_prob, S, O->po_vert, O->cur.mp_po, O->po_obs, O->po_is2, O->cur.mp_valid, O->cur.outlier, t->ob_chi2, t->ob_state,
                       t->ob_cedge, O->po_pose, O->po_result, nullptr, st);
    mark(TS_OBJ_CFSE3);
    psk_ob_finish(O, t->step, st); mark(TS_OBJ_GLUE);
  }
  t->ev_count = e;
  if (tm) t->timed++;
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
    for (int i = 0; i < kTrkMaxEvents; i++) hipEventCreate(&t->ev[r][i]);
  if (cfg->max_objects > 0) {
    // the object half: its own cv::ORB detector (Frame.cc:2625: cv::ORB::create(1000, 1.2, 8, 19)) and the arrays of objtrack_plan.h
    if (cfg->max_objects > OB_MAXK) { ps_tracker_destroy(t); return ps_set_error(PS_ERR_INVALID, "max_objects: at most %d detections per frame", OB_MAXK); }
    if (cfg->max_map_objects < 0 || cfg->max_map_objects > OB_MAXM) { ps_tracker_destroy(t); return ps_set_error(PS_ERR_INVALID, "max_map_objects: 0 (= 8) .. %d", OB_MAXM); }
    // ob_masks keeps one row x 3 planes of the padded width in LDS (psk_ob_masks: 3 WP + 64 bytes; at most the 64 KB a kernel gets without asking for more)
    if (3 * (size_t)((cfg->width + 255) & ~255) + 64 > 64 * 1024) {
      ps_tracker_destroy(t);
      return ps_set_error(PS_ERR_CAPACITY, "ps_tracker_create: the object chain serves images up to 21760 pixels wide (%d asked)", cfg->width);
    }
    rc = ps_cvorb_create(1000, 1.2f, 8, 19, 20, cfg->device, &t->cvorb);
    if (rc == PS_OK) {
      // plan the detector for 2 S images of this size (one untimed batch over zero masks: allocations happen here, not in the first step)
      uint8_t* z = nullptr;
      const size_t img_bytes = (size_t)cfg->width * cfg->height;
      if (hipMalloc(&z, 2 * (size_t)A.S * img_bytes) != hipSuccess) rc = ps_set_error(PS_ERR_HIP, "hipMalloc failed");
      else {
        hipMemsetAsync(z, 0, 2 * (size_t)A.S * img_bytes, t->stream);
        rc = ps_cvorb_detect_batch_device(t->cvorb, z, z, 2 * A.S, cfg->width, cfg->height, cfg->width, img_bytes, cfg->width, img_bytes, t->stream);
        hipStreamSynchronize(t->stream);
        hipFree(z);
      }
    }
    if (rc != PS_OK) { ps_tracker_destroy(t); return rc; }
    ObArrays& O = t->OA;
    memset(&O, 0, sizeof(O));
    O.S = A.S; O.K = cfg->max_objects; O.M = cfg->max_map_objects > 0 ? cfg->max_map_objects : 8; O.max_steps = cfg->max_steps;
    int32_t ccap = 0;
    ps_cvorb_batch_device_outputs(t->cvorb, nullptr, nullptr, nullptr, nullptr, &ccap);
    O.OC = ccap; O.LC = 1024;
    ObCam& OC = O.cam;
    OC.fx = C.fx; OC.fy = C.fy; OC.cx = C.cx; OC.cy = C.cy; OC.mbf = C.mbf; OC.mb = C.mb; OC.th_depth = C.th_depth; OC.gw_inv = C.gw_inv; OC.gh_inv = C.gh_inv;
    OC.log_sf = C.log_sf; OC.inv_fx = C.inv_fx; OC.inv_fy = C.inv_fy; OC.w = C.w; OC.h = C.h; OC.nlevels = C.nlevels;
    for (int l = 0; l < 8; l++) { OC.sf[l] = C.sf[l]; OC.inv_sigma2[l] = C.inv_sigma2[l]; }
    const size_t ob_bytes = carve_obj(t, nullptr);
    e = hipMalloc(&t->d_obj, ob_bytes);
    if (e != hipSuccess) { ps_tracker_destroy(t); return ps_set_error(PS_ERR_HIP, "hipMalloc(%zu): %s", ob_bytes, hipGetErrorString(e)); }
    t->d_obj_bytes = ob_bytes;
    hipMemsetAsync(t->d_obj, 0, ob_bytes, t->stream);
    hipStreamSynchronize(t->stream);   // the tables below are written with synchronous copies: not before the clear has run
    carve_obj(t, t->d_obj);
    // every MapObject slot free
    std::vector<ObMapObject> mo((size_t)O.S * O.M);
    memset(mo.data(), 0, mo.size() * sizeof(ObMapObject));
    for (ObMapObject& m : mo) m.id = -1;
    hipMemcpy(O.mobj, mo.data(), mo.size() * sizeof(ObMapObject), hipMemcpyHostToDevice);
    O.cam_traj = A.traj; O.cam_stats = (const int32_t*)A.stats; O.cam_stat_words = (int32_t)(sizeof(TrkStat) / 4);
    const char* ov = getenv("PS_TRK_OVERLAP");
    t->overlap = ov ? atoi(ov) != 0 : A.S <= 32;
    if (t->overlap) {
      int least = 0, greatest = 0;
      hipDeviceGetStreamPriorityRange(&least, &greatest);          // the camera chain is the critical path: the object features fill in behind it
      if (hipStreamCreateWithPriority(&t->stream2, hipStreamNonBlocking, least) != hipSuccess || hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t->ev_orb, hipEventDisableTiming) != hipSuccess) { ps_tracker_destroy(t); return ps_set_error(PS_ERR_HIP, "ps_tracker_create: second stream"); }
      for (int r = 0; r < ps_tracker::RING; r++) { hipEventCreate(&t->ev_obj[r][0]); hipEventCreate(&t->ev_obj[r][1]); }
    }
  }
  if (hipStreamSynchronize(t->stream) != hipSuccess) { ps_tracker_destroy(t); return ps_set_error(PS_ERR_HIP, "tracker initialisation failed"); }
  *out = t;
  return PS_OK;
}

void ps_tracker_destroy(ps_tracker* t) {
  if (!t) return;
  hipSetDevice(t->cfg.device);
  if (t->stream) hipStreamSynchronize(t->stream);
  for (int r = 0; r < ps_tracker::RING; r++)
    for (int i = 0; i < kTrkMaxEvents; i++) if (t->ev[r][i]) hipEventDestroy(t->ev[r][i]);
  if (t->stream2) { hipStreamSynchronize(t->stream2); hipStreamDestroy(t->stream2); }
  if (t->ev_fork) hipEventDestroy(t->ev_fork);
  if (t->ev_join) hipEventDestroy(t->ev_join);
  if (t->ev_orb) hipEventDestroy(t->ev_orb);
  for (int r = 0; r < ps_tracker::RING; r++) for (int i = 0; i < 2; i++) if (t->ev_obj[r][i]) hipEventDestroy(t->ev_obj[r][i]);
  if (t->d_buf) hipFree(t->d_buf);
  if (t->d_obj) hipFree(t->d_obj);
  if (t->cvorb) ps_cvorb_destroy(t->cvorb);
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
  return queue_chain(t, d_imgs, stride, image_pitch, nullptr, 0, 0, nullptr);
}

int ps_tracker_step_slot_device(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch, const uint8_t* d_masks, int mask_stride,
                                size_t mask_pitch, const ps_detection* d_dets) {
  if (!t || !d_imgs || !d_masks || !d_dets) return ps_set_error(PS_ERR_INVALID, "ps_tracker_step_slot_device: null argument");
  if (!t->cvorb) return ps_set_error(PS_ERR_INVALID, "the tracker was created without objects (max_objects = 0)");
  if (mask_stride < t->cfg.width) return ps_set_error(PS_ERR_INVALID, "mask stride < width");
  if (t->step >= t->A.max_steps) return ps_set_error(PS_ERR_CAPACITY, "the tracker was created for %d steps", t->A.max_steps);
  PS_HIP(hipSetDevice(t->cfg.device));
  if (t->timing) PS_HIP(hipEventRecord(t->ev[t->timed % ps_tracker::RING][0], t->stream));
  int rc = PS_OK;
  if (t->overlap && (rc = overlap_head_before_orb(t, d_imgs, stride, image_pitch, d_masks, mask_stride, mask_pitch, d_dets)) != PS_OK) return rc;
  rc = ps_orb_extract_batch_device(t->orb, d_imgs, 2 * t->A.S, t->cfg.width, t->cfg.height, stride, image_pitch, nullptr);
  if (rc != PS_OK) return rc;
  if (t->overlap && (rc = overlap_head_after_orb(t)) != PS_OK) return rc;
  return queue_chain(t, d_imgs, stride, image_pitch, d_masks, mask_stride, mask_pitch, d_dets);
}

int ps_tracker_fetch_objects(ps_tracker* t, int first_step, int nsteps, ps_object_stat* out) {
  if (!t || !t->cvorb || !out || first_step < 0 || nsteps < 0 || first_step + nsteps > t->step)
    return ps_set_error(PS_ERR_INVALID, "ps_tracker_fetch_objects: steps [%d, %d) of %d", first_step, first_step + nsteps, t ? t->step : 0);
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  static_assert(sizeof(ps_object_stat) == sizeof(ObStat), "ps_object_stat layout");
  static_assert(sizeof(ps_detection) == sizeof(ObDet), "ps_detection layout");
  const size_t per = (size_t)t->OA.S * t->OA.K;
  if (nsteps) PS_HIP(hipMemcpy(out, t->OA.stats + (size_t)first_step * per, (size_t)nsteps * per * sizeof(ObStat), hipMemcpyDeviceToHost));
  // the detector's limits, the object searches' candidate store and the MapObject table are part of the result's validity: their
  // flags are kept per sequence over ALL queued steps (ob_begin / ob_finish), until ps_tracker_reset
  std::vector<int32_t> ovf(t->OA.S, 0), sov(t->OA.S, 0), drop(t->OA.S, 0);
  PS_HIP(hipMemcpy(ovf.data(), t->OA.det_overflow, ovf.size() * 4, hipMemcpyDeviceToHost));
  PS_HIP(hipMemcpy(sov.data(), t->OA.search_overflow, sov.size() * 4, hipMemcpyDeviceToHost));
  PS_HIP(hipMemcpy(drop.data(), t->OA.dropped, drop.size() * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < ovf.size(); i++)
    if (ovf[i]) return ps_set_error(PS_ERR_CAPACITY, "sequence %zu: the object detector exceeded its per-level / per-image keypoint capacity in %d step(s) since the last reset", i, ovf[i]);
  for (size_t i = 0; i < sov.size(); i++)
    if (sov[i]) return ps_set_error(PS_ERR_CAPACITY, "sequence %zu: %d object search windows held more than %d candidates since the last reset", i, sov[i], PS_PJ_CAP);
  for (size_t i = 0; i < drop.size(); i++)
    if (drop[i]) return ps_set_error(PS_ERR_CAPACITY, "sequence %zu: more than %d objects over the sequence (%d detections ignored; ps_tracker_config.max_map_objects)", i, t->OA.M, drop[i]);
  return PS_OK;
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
  return queue_chain(t, nullptr, 0, 0, nullptr, 0, 0, nullptr);
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
  return PS_OK;
}

// test hook: pretends that `count` search windows of sequence `seq` overflowed in the step that is queued next
int ps_tracker_debug_set_overflow(ps_tracker* t, int seq, int count) {
  if (!t || seq < 0 || seq >= t->A.S) return ps_set_error(PS_ERR_INVALID, "bad argument");
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  PS_HIP(hipMemcpy(t->d_overflow + seq, &count, 4, hipMemcpyHostToDevice));
  return PS_OK;
}

int ps_tracker_reset(ps_tracker* t) {
  if (!t) return ps_set_error(PS_ERR_INVALID, "null handle");
  PS_HIP(hipSetDevice(t->cfg.device));
  PS_HIP(hipStreamSynchronize(t->stream));
  PS_HIP(hipMemsetAsync(t->A.seq, 0, sizeof(TrkSeq) * t->A.S, t->stream));
  PS_HIP(hipMemsetAsync(t->d_overflow, 0, sizeof(int32_t) * t->A.S, t->stream));
  if (t->cvorb) {
    std::vector<ObMapObject> mo((size_t)t->OA.S * t->OA.M);
    memset(mo.data(), 0, mo.size() * sizeof(ObMapObject));
    for (ObMapObject& m : mo) m.id = -1;
    PS_HIP(hipMemcpy(t->OA.mobj, mo.data(), mo.size() * sizeof(ObMapObject), hipMemcpyHostToDevice));
    PS_HIP(hipMemsetAsync(t->OA.last.ndet, 0, sizeof(int32_t) * t->OA.S, t->stream));
    PS_HIP(hipMemsetAsync(t->OA.dropped, 0, sizeof(int32_t) * t->OA.S, t->stream));
    PS_HIP(hipMemsetAsync(t->OA.det_overflow, 0, sizeof(int32_t) * t->OA.S, t->stream));
    PS_HIP(hipMemsetAsync(t->OA.search_overflow, 0, sizeof(int32_t) * t->OA.S, t->stream));
    PS_HIP(hipMemsetAsync(t->OA.pj_overflow, 0, sizeof(int32_t) * t->OA.S * t->OA.K, t->stream));
  }
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
  for (int r = 0; r < cnt; r++)
    for (int i = 1; i < t->ev_count; i++) {
      float v = 0;
      PS_HIP(hipEventElapsedTime(&v, t->ev[r][i - 1], t->ev[r][i]));
      acc[t->ev_stage[i]] += v;
    }
  if (t->overlap && t->cvorb)
    for (int r = 0; r < cnt; r++) {
      float v = 0;
      if (hipEventElapsedTime(&v, t->ev_obj[r][0], t->ev_obj[r][1]) == hipSuccess) acc[TS_OBJ_FEATURES] += v;   // on the second stream, beside the camera chain
    }
  *n = TS_COUNT;
  for (int i = 0; i < TS_COUNT && i < cap; i++) {
    if (names) names[i] = kTrkStage[i];
    if (ms) ms[i] = cnt ? (float)(acc[i] / cnt) : 0.f;
  }
  return PS_OK;
}

}  // extern "C"
