// The tracking thread's data flow around the hot path in C++ (the same slice as pointslot_amd/tracker.py, which documents
// the mapping to /root/reference/src/Tracking.cc:2840-3160,1260-1286 and src/Frame.cc:1686-1743,2505-2519):
//   Frame::Frame (ExtractORB x2 + ComputeStereoMatches), StereoInitialization, UpdateLastFrame (localisation mode),
//   TrackWithMotionModel, SearchLocalPoints / TrackLocalMap, the constant-velocity model.
// Host code only; every heavy step is one call into the C-ABI.  float arithmetic where the reference uses cv::Mat CV_32F.
//
// OdoSequence holds one sequence's tracking state and runs the host side of Tracking::Track as a small state machine: it
// prepares the next device request (a SearchByProjection problem or a PoseOptimization problem) and continues once the
// request has been served.  Two drivers serve the requests:
//   StereoOdometry       one sequence, one frame at a time, the reference's call structure (two extractor objects on two
//                        threads, then one C-ABI call per step)
//   StereoOdometryBatch  many independent sequences in lockstep: one batched extraction + stereo matching for all frames
//                        of the step, then every round's search / pose problems of all sequences in ONE C-ABI call each
//                        (BASELINE config 4: independent sequences are the unit of parallelism)
// Both produce the same trajectories: a problem's result does not depend on what else is in its batch.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"

namespace ORB_SLAM2 {

struct OdoFrame {
  int N = 0;
  std::vector<pscv::KeyPoint> mvKeys;
  pscv::Mat mDescriptors;
  std::vector<float> x, y, angle, mvuRight, mvDepth;
  std::vector<int32_t> octave, cell_off, cell_idx;
  std::vector<float> mp_xw;             // [N][3] world position of mvpMapPoints[i]
  std::vector<uint8_t> mp_valid, mp_observed, outlier;
  std::vector<int32_t> mp_id;
  float tcw[16];
  bool has_pose = false;
};

// intrinsics and the extractor tables every sequence of a rig shares
struct OdoCamera {
  float fx, fy, cx, cy, mbf, mb, mThDepth, gwInv, ghInv, logSf;
  int w, h;
  std::vector<float> sf, invSigma2;
  static const int GRID_COLS = 64, GRID_ROWS = 48;
  OdoCamera(float fx_, float fy_, float cx_, float cy_, float bf, int width, int height, float thDepth, const std::vector<float>& scaleFactors,
            const std::vector<float>& invLevelSigma2)
      : fx(fx_), fy(fy_), cx(cx_), cy(cy_), mbf(bf), w(width), h(height), sf(scaleFactors), invSigma2(invLevelSigma2) {
    mb = mbf / fx;
    mThDepth = mbf * thDepth / fx;                                                       // Tracking.cc:402
    gwInv = (float)GRID_COLS / (float)width; ghInv = (float)GRID_ROWS / (float)height;   // Frame.cc:1636-1640, no distortion
    logSf = std::log(sf[1]);
  }
};

class OdoSequence {
 public:
  enum State { NOT_INITIALIZED, OK, LOST };
  enum Request { NONE, SEARCH, POSE };

  explicit OdoSequence(const OdoCamera* camera) : cam(camera) {}

  State state = NOT_INITIALIZED;
  std::vector<std::vector<float>> trajectory;   // Tcw (16 floats) per frame, empty when lost / not initialised
  int lastMatches = 0, lastMapMatches = 0, lastLocalInliers = 0;
  bool lastFrameTracked = false;

  ps_proj_problem proj;     // the pending SearchByProjection problem (Request SEARCH)
  ps_pose_problem posep;    // the pending PoseOptimization problem (Request POSE)

  // Tracking::Track up to the first device request.  F holds the extraction results (mvKeys, mDescriptors, mvuRight, mvDepth).
  Request begin(std::unique_ptr<OdoFrame> frame) {
    F = std::move(frame);
    finishFrame(*F);
    lastFrameTracked = false;
    if (state == NOT_INITIALIZED) {
      if (initialize(*F)) { last = std::move(F); haveVelocity = false; trajectory.push_back(pose(*last)); lastFrameTracked = true; return NONE; }
      trajectory.push_back({});
      F.reset();
      return NONE;
    }
    if (!haveVelocity) { setIdentity(velocity); haveVelocity = true; }   // see tracker.py: no vocabulary for TrackReferenceKeyFrame
    // Tracking::TrackWithMotionModel
    OdoFrame& L = *last;
    updateLastFrame();
    mul4(velocity, L.tcw, F->tcw); F->has_pose = true;
    qvalid.assign(L.N, 0); qobs.assign(L.N, 1); occupied.assign(F->N, 0);
    for (int i = 0; i < L.N; i++) qvalid[i] = (L.mp_valid[i] && !L.outlier[i]) ? 1 : 0;
    match.assign(std::max(F->N, 1), -1);
    th = 7.f;
    fillMotionModelSearch();
    stage = MM_SEARCH;
    return SEARCH;
  }

  // continues after the pending request has been served (proj.nmatches / posep.* are filled)
  Request advance() {
    switch (stage) {
      case MM_SEARCH: {
        if (proj.nmatches < 20 && th == 7.f) { th = 14.f; fillMotionModelSearch(); return SEARCH; }   // th, then 2 * th
        if (proj.nmatches < 20) return fail();
        OdoFrame& L = *last;
        for (int j = 0; j < F->N; j++) {
          const int i = match[j];
          F->mp_valid[j] = i >= 0;
          if (i >= 0) { std::memcpy(&F->mp_xw[3 * (size_t)j], &L.mp_xw[3 * (size_t)i], 12); F->mp_observed[j] = L.mp_observed[i]; F->mp_id[j] = L.mp_id[i]; }
        }
        fillPose();
        stage = MM_POSE;
        return POSE;
      }
      case MM_POSE: {
        takePose();
        int nmatches = 0, nmatchesMap = 0;
        for (int i = 0; i < F->N; i++) {                                 // discard outliers (Tracking.cc:3062-3082)
          if (!F->mp_valid[i]) continue;
          if (F->outlier[i]) { F->mp_valid[i] = 0; F->outlier[i] = 0; continue; }
          nmatches++;
          if (F->mp_observed[i]) nmatchesMap++;
        }
        lastMatches = nmatches; lastMapMatches = nmatchesMap;
        mbVO = nmatchesMap < 10;
        if (!(nmatches > 20)) return fail();
        if (mbVO) return finish();
        // SearchLocalPoints + TrackLocalMap
        if (fillLocalMapSearch() > 0) { stage = LM_SEARCH; return SEARCH; }
        fillPose();
        stage = LM_POSE;
        return POSE;
      }
      case LM_SEARCH: {
        for (int j = 0; j < F->N; j++) {
          const int i = match[j];
          if (i < 0) continue;
          F->mp_valid[j] = 1; std::memcpy(&F->mp_xw[3 * (size_t)j], &lm_xw[3 * (size_t)i], 12); F->mp_observed[j] = 1; F->mp_id[j] = i;
        }
        fillPose();
        stage = LM_POSE;
        return POSE;
      }
      case LM_POSE: {
        takePose();
        int inl = 0;
        for (int i = 0; i < F->N; i++) {
          if (!F->mp_valid[i]) continue;
          if (F->outlier[i]) F->mp_valid[i] = 0;                          // stereo: outliers lose their map point (Tracking.cc:3141-3142)
          else inl++;
        }
        lastLocalInliers = inl;
        if (inl < 30) return fail();
        return finish();
      }
    }
    return NONE;
  }

 private:
  enum Stage { MM_SEARCH, MM_POSE, LM_SEARCH, LM_POSE };
  const OdoCamera* cam;
  Stage stage = MM_SEARCH;
  std::unique_ptr<OdoFrame> F, last;
  float velocity[16];
  bool haveVelocity = false, mbVO = false;
  float th = 7.f;
  // the initial keyframe's map points (the local map of this slice)
  std::vector<float> lm_xw, lm_normal, lm_maxd, lm_mind;
  std::vector<uint8_t> lm_desc;
  // buffers the pending request points into
  std::vector<uint8_t> qvalid, qobs, occupied;
  std::vector<int32_t> match, minl, maxl;
  std::vector<float> qu, qv, qur, rad, obs, is2;
  int nvalidPose = 0;

  static void setIdentity(float* m) { std::memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }
  static void mul4(const float* a, const float* b, float* o) {
    float t[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        float acc = 0;
        for (int k = 0; k < 4; k++) acc += a[4 * r + k] * b[4 * k + c];
        t[4 * r + c] = acc;
      }
    std::memcpy(o, t, 64);
  }
  static std::vector<float> pose(const OdoFrame& F) { return std::vector<float>(F.tcw, F.tcw + 16); }

  Request fail() { state = LOST; haveVelocity = false; trajectory.push_back({}); F.reset(); return NONE; }

  Request finish() {
    // mVelocity = Tcw * LastTwc (Tracking.cc:1260-1270)
    float lastTwc[16];
    setIdentity(lastTwc);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) lastTwc[4 * r + c] = last->tcw[4 * c + r];
    for (int r = 0; r < 3; r++) {
      float acc = 0;
      for (int c = 0; c < 3; c++) acc += last->tcw[4 * c + r] * last->tcw[4 * c + 3];
      lastTwc[4 * r + 3] = -acc;
    }
    mul4(F->tcw, lastTwc, velocity);
    for (int i = 0; i < F->N; i++)                     // clean VO matches (Tracking.cc:1274-1286)
      if (F->mp_valid[i] && !F->mp_observed[i]) { F->mp_valid[i] = 0; F->outlier[i] = 0; }
    last = std::move(F);
    trajectory.push_back(pose(*last));
    lastFrameTracked = true;
    state = OK;                                        // `if (bOK) mState = OK;` - also after a frame that was lost
    return NONE;
  }

  // the part of Frame::Frame after the extractors: per-keypoint arrays, AssignFeaturesToGrid, empty map-point slots
  void finishFrame(OdoFrame& Fr) const {
    Fr.N = (int)Fr.mvKeys.size();
    Fr.x.resize(Fr.N); Fr.y.resize(Fr.N); Fr.angle.resize(Fr.N); Fr.octave.resize(Fr.N);
    for (int i = 0; i < Fr.N; i++) { Fr.x[i] = Fr.mvKeys[i].pt.x; Fr.y[i] = Fr.mvKeys[i].pt.y; Fr.angle[i] = Fr.mvKeys[i].angle; Fr.octave[i] = Fr.mvKeys[i].octave; }
    // Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:1636-1656, 2027-2037) as CSR, cell = ix * 48 + iy
    const int GC = OdoCamera::GRID_COLS, GR = OdoCamera::GRID_ROWS;
    std::vector<int> cell(Fr.N, -1);
    Fr.cell_off.assign(GC * GR + 1, 0);
    for (int i = 0; i < Fr.N; i++) {
      const int px = (int)std::round((Fr.x[i] - 0.f) * cam->gwInv), py = (int)std::round((Fr.y[i] - 0.f) * cam->ghInv);
      if (px < 0 || px >= GC || py < 0 || py >= GR) continue;
      cell[i] = px * GR + py;
      Fr.cell_off[cell[i] + 1]++;
    }
    for (int c = 0; c < GC * GR; c++) Fr.cell_off[c + 1] += Fr.cell_off[c];
    Fr.cell_idx.assign(Fr.cell_off.back(), 0);
    std::vector<int> fill(Fr.cell_off.begin(), Fr.cell_off.end() - 1);
    for (int i = 0; i < Fr.N; i++) if (cell[i] >= 0) Fr.cell_idx[fill[cell[i]]++] = i;
    Fr.mp_xw.assign((size_t)Fr.N * 3, 0.f); Fr.mp_valid.assign(Fr.N, 0); Fr.mp_observed.assign(Fr.N, 0); Fr.outlier.assign(Fr.N, 0);
    Fr.mp_id.assign(Fr.N, -1);
  }

  // Frame::UnprojectStereo (Frame.cc:2505-2519)
  void unproject(const OdoFrame& Fr, int i, float* X) const {
    const float z = Fr.mvDepth[i];
    const float xc = (Fr.x[i] - cam->cx) * z * (1.f / cam->fx), yc = (Fr.y[i] - cam->cy) * z * (1.f / cam->fy);
    const float* T = Fr.tcw;
    float Ow[3];
    for (int r = 0; r < 3; r++) Ow[r] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
    for (int r = 0; r < 3; r++) X[r] = (T[r] * xc + T[4 + r] * yc + T[8 + r] * z) + Ow[r];   // Rwc * x3Dc + mOw
  }

  bool initialize(OdoFrame& Fr) {                                    // Tracking::StereoInitialization
    if (Fr.N <= 500) return false;
    setIdentity(Fr.tcw); Fr.has_pose = true;
    int id = 0;
    for (int i = 0; i < Fr.N; i++) {
      if (!(Fr.mvDepth[i] > 0)) continue;
      unproject(Fr, i, &Fr.mp_xw[3 * (size_t)i]);
      Fr.mp_valid[i] = 1; Fr.mp_observed[i] = 1; Fr.mp_id[i] = id++;
      // MapPoint::UpdateNormalAndDepth with one observation (MapPoint.cc:470-497): camera centre at the origin
      const float* P = &Fr.mp_xw[3 * (size_t)i];
      const float dist = std::sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
      const float maxd = dist * cam->sf[Fr.octave[i]];
      for (int c = 0; c < 3; c++) { lm_xw.push_back(P[c]); lm_normal.push_back(P[c] / dist); }
      lm_maxd.push_back(maxd); lm_mind.push_back(maxd / cam->sf.back());
      lm_desc.insert(lm_desc.end(), Fr.mDescriptors.ptr<uint8_t>(i), Fr.mDescriptors.ptr<uint8_t>(i) + 32);
    }
    state = OK;
    return true;
  }

  void updateLastFrame() {                                          // Tracking::UpdateLastFrame, localisation mode
    OdoFrame& L = *last;
    std::vector<std::pair<float, int>> vDepthIdx;
    for (int i = 0; i < L.N; i++) if (L.mvDepth[i] > 0) vDepthIdx.push_back(std::make_pair(L.mvDepth[i], i));
    if (vDepthIdx.empty()) return;
    std::sort(vDepthIdx.begin(), vDepthIdx.end());
    int nPoints = 0;
    for (size_t j = 0; j < vDepthIdx.size(); j++) {
      const int i = vDepthIdx[j].second;
      if (!L.mp_valid[i] || !L.mp_observed[i]) {
        unproject(L, i, &L.mp_xw[3 * (size_t)i]);
        L.mp_valid[i] = 1; L.mp_observed[i] = 0; L.mp_id[i] = -1;
      }
      nPoints++;
      if (vDepthIdx[j].first > 2 * cam->mThDepth && nPoints > 100) break;
    }
  }

  void fillTrain(ps_proj_train& t) const {
    const OdoFrame& Fr = *F;
    t.n = Fr.N; t.x = Fr.x.data(); t.y = Fr.y.data(); t.octave = Fr.octave.data(); t.angle = Fr.angle.data(); t.u_right = Fr.mvuRight.data();
    t.desc = Fr.mDescriptors.data; t.occupied = occupied.data(); t.in_bbox = nullptr; t.cell_off = Fr.cell_off.data(); t.cell_idx = Fr.cell_idx.data();
    t.min_x = 0.f; t.min_y = 0.f; t.grid_w_inv = cam->gwInv; t.grid_h_inv = cam->ghInv;
  }

  void fillMotionModelSearch() {      // matcher(0.9, true).SearchByProjection(mCurrentFrame, mLastFrame, th, false) (Tracking.cc:3030-3048)
    const OdoFrame& L = *last;
    proj = ps_proj_problem{};
    fillTrain(proj.train);
    proj.nq = L.N; proj.q_valid = qvalid.data(); proj.q_desc = L.mDescriptors.data; proj.q_observed = qobs.data(); proj.q_angle = L.angle.data();
    proj.q_xw = L.mp_xw.data(); proj.q_octave = L.octave.data(); proj.mono = 0;
    std::memcpy(proj.tcw, F->tcw, 64); std::memcpy(proj.tlw, L.tcw, 64);
    proj.fx = cam->fx; proj.fy = cam->fy; proj.cx = cam->cx; proj.cy = cam->cy; proj.mbf = cam->mbf; proj.mb = cam->mb;
    proj.bounds[0] = 0.f; proj.bounds[1] = (float)cam->w; proj.bounds[2] = 0.f; proj.bounds[3] = (float)cam->h;
    for (int l = 0; l < 8; l++) proj.scale_factors[l] = l < (int)cam->sf.size() ? cam->sf[l] : 1.f;
    proj.th = th; proj.match_of_train = match.data();
    ORBmatcher::ConfigureProjectionFrame(proj, 0.9f, true);
  }

  int fillLocalMapSearch() {          // SearchLocalPoints: isInFrustum + matcher(0.8).SearchByProjection(mCurrentFrame, points, th = 1)
    OdoFrame& Fr = *F;
    const int n = (int)lm_maxd.size();
    std::vector<uint8_t> already(n, 0);
    qvalid.assign(n, 0); qobs.assign(n, 1); occupied.assign(Fr.N, 0);
    for (int i = 0; i < Fr.N; i++) {
      if (Fr.mp_id[i] >= 0) already[Fr.mp_id[i]] = 1;   // mnLastFrameSeen: matched points and the outliers just discarded (Tracking.cc:3071-3075)
      occupied[i] = (Fr.mp_valid[i] && Fr.mp_observed[i]) ? 1 : 0;
    }
    qu.assign(n, 0.f); qv.assign(n, 0.f); qur.assign(n, 0.f); rad.assign(n, 0.f);
    minl.assign(n, 0); maxl.assign(n, 0);
    const float* T = Fr.tcw;
    const float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
    float Ow[3];
    for (int r = 0; r < 3; r++) Ow[r] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
    int nto = 0;
    for (int i = 0; i < n; i++) {                                   // Frame::isInFrustum (Frame.cc:1686-1743), viewingCosLimit 0.5
      if (already[i]) continue;
      const float* P = &lm_xw[3 * (size_t)i];
      const float PcX = T[0] * P[0] + T[1] * P[1] + T[2] * P[2] + T[3], PcY = T[4] * P[0] + T[5] * P[1] + T[6] * P[2] + T[7],
                  PcZ = T[8] * P[0] + T[9] * P[1] + T[10] * P[2] + T[11];
      if (PcZ < 0.0f) continue;
      const float invz = 1.0f / PcZ, u = fx * PcX * invz + cx, v = fy * PcY * invz + cy;
      if (u < 0 || u > (float)cam->w || v < 0 || v > (float)cam->h) continue;
      const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
      const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      if (dist < 0.8f * lm_mind[i] || dist > 1.2f * lm_maxd[i]) continue;
      const float viewCos = (PO[0] * lm_normal[3 * (size_t)i] + PO[1] * lm_normal[3 * (size_t)i + 1] + PO[2] * lm_normal[3 * (size_t)i + 2]) / dist;
      if (viewCos < 0.5f) continue;
      int level = (int)std::ceil(std::log(lm_maxd[i] / dist) / cam->logSf);   // MapPoint::PredictScale
      level = level < 0 ? 0 : (level >= (int)cam->sf.size() ? (int)cam->sf.size() - 1 : level);
      const float r = ORBmatcher::RadiusByViewingCos(viewCos);      // th = 1: no factor (ORBmatcher.cc:74-84)
      qvalid[i] = 1; qu[i] = u; qv[i] = v; qur[i] = u - cam->mbf * invz; rad[i] = r * cam->sf[level]; minl[i] = level - 1; maxl[i] = level;
      nto++;
    }
    if (nto == 0) return 0;
    match.assign(std::max(Fr.N, 1), -1);
    proj = ps_proj_problem{};
    fillTrain(proj.train);
    proj.nq = n; proj.q_valid = qvalid.data(); proj.q_desc = lm_desc.data(); proj.q_observed = qobs.data();
    proj.q_u = qu.data(); proj.q_v = qv.data(); proj.q_ur = qur.data(); proj.q_radius = rad.data(); proj.q_radius_er = rad.data();
    proj.q_min_level = minl.data(); proj.q_max_level = maxl.data();
    for (int l = 0; l < 8; l++) proj.scale_factors[l] = l < (int)cam->sf.size() ? cam->sf[l] : 1.f;
    proj.th = 1.f; proj.match_of_train = match.data();
    ORBmatcher::ConfigureProjectionPoints(proj, 0.8f);
    return nto;
  }

  void fillPose() {                                                 // Optimizer::PoseOptimization(&mCurrentFrame)
    OdoFrame& Fr = *F;
    obs.resize((size_t)Fr.N * 3); is2.resize(Fr.N);
    nvalidPose = 0;
    for (int i = 0; i < Fr.N; i++) {
      obs[3 * (size_t)i] = Fr.x[i]; obs[3 * (size_t)i + 1] = Fr.y[i]; obs[3 * (size_t)i + 2] = Fr.mvuRight[i];
      is2[i] = cam->invSigma2[Fr.octave[i]];
      nvalidPose += Fr.mp_valid[i] ? 1 : 0;
    }
    posep = ps_pose_problem{};
    posep.n = Fr.N; posep.xw = Fr.mp_xw.data(); posep.obs = obs.data(); posep.inv_sigma2 = is2.data(); posep.valid = Fr.mp_valid.data();
    posep.fx = cam->fx; posep.fy = cam->fy; posep.cx = cam->cx; posep.cy = cam->cy; posep.bf = cam->mbf;
    std::memcpy(posep.tcw, Fr.tcw, 64);
    posep.outlier = Fr.outlier.data();
  }
  void takePose() {
    if (nvalidPose >= 15) std::memcpy(F->tcw, posep.tcw, 64);       // Optimizer.cc:376-377
  }
};

// ---- one sequence, the reference's call structure ----
class StereoOdometry {
 public:
  typedef OdoSequence::State State;
  static const State NOT_INITIALIZED = OdoSequence::NOT_INITIALIZED, OK = OdoSequence::OK, LOST = OdoSequence::LOST;

  StereoOdometry(float fx, float fy, float cx, float cy, float bf, int width, int height, float thDepth = 35.f, int nFeatures = 2000,
                 float scale = 1.2f, int nLevels = 8, int iniTh = 20, int minTh = 5)
      : left(nFeatures, scale, nLevels, iniTh, minTh), right(nFeatures, scale, nLevels, iniTh, minTh), matcher(0.9f, true),
        cam(fx, fy, cx, cy, bf, width, height, thDepth, left.GetScaleFactors(), left.GetInverseScaleSigmaSquares()), seq(&cam),
        state(seq.state), trajectory(seq.trajectory), lastMatches(seq.lastMatches), lastMapMatches(seq.lastMapMatches),
        lastLocalInliers(seq.lastLocalInliers) {
    left.mbDownloadPyramid = false; right.mbDownloadPyramid = false;
  }

  // Where a frame's wall time goes ("marshaling to SoA every call must be measured and reported separately from kernel
  // time"), accumulated over Track calls: wall seconds inside the C-ABI calls by kind, the kernels' own time inside the matcher / optimiser
  // calls (HIP events around the launches, read back through ps_*_last_kernel_ms), and everything else = the host side of Tracking::Track
  // (marshalling into the problem structs, match application, grids).
  struct Split {
    double extract = 0, stereo = 0, search = 0, pose = 0, host = 0;     // wall seconds
    double searchKernelMs = 0, poseKernelMs = 0;
    long searchCalls = 0, poseCalls = 0, frames = 0;
    void clear() { *this = Split(); }
  } split;

  // Tracking::Track for one stereo frame; returns true when the frame has a pose
  bool Track(const pscv::Mat& imLeft, const pscv::Mat& imRight) {
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = clk::now();
    std::unique_ptr<OdoFrame> F(new OdoFrame);
    std::vector<pscv::KeyPoint> keysR;
    pscv::Mat descR;
    // two threads, one extractor each, as the reference does (Frame.cc:709-710): the handles own separate streams, so the two
    // single-image pipelines overlap on the GPU
    std::thread threadLeft([&]() { left(imLeft, pscv::Mat(), F->mvKeys, F->mDescriptors); });
    std::thread threadRight([&]() { right(imRight, pscv::Mat(), keysR, descR); });
    threadLeft.join();
    threadRight.join();
    const auto t1 = clk::now();
    const int N = (int)F->mvKeys.size();
    F->mvuRight.assign(N, -1.f); F->mvDepth.assign(N, -1.f);
    int n = 0;
    if (N > 0 && ps_orb_stereo_match_pair(left.handle(), right.handle(), cam.mb, cam.mbf, F->mvuRight.data(), F->mvDepth.data(), N, &n) != PS_OK)
      throw std::runtime_error(ps_last_error());                                         // Frame::ComputeStereoMatches
    const auto t2 = clk::now();
    split.extract += secs(t0, t1); split.stereo += secs(t1, t2);
    double inCalls = 0;
    OdoSequence::Request rq = seq.begin(std::move(F));
    while (rq != OdoSequence::NONE) {
      const auto a = clk::now();
      float kms = 0;
      if (rq == OdoSequence::SEARCH) {
        matcher.SearchByProjectionBatch(&seq.proj, 1);
        const double d = secs(a, clk::now());
        ps_matcher_last_kernel_ms(matcher.handle(), &kms);
        split.search += d; split.searchKernelMs += kms; split.searchCalls++; inCalls += d;
      } else {
        Optimizer::PoseOptimization(&seq.posep);
        const double d = secs(a, clk::now());
        ps_optimizer_last_kernel_ms(Optimizer::handle(), &kms);
        split.pose += d; split.poseKernelMs += kms; split.poseCalls++; inCalls += d;
      }
      rq = seq.advance();
    }
    split.host += secs(t2, clk::now()) - inCalls;
    split.frames++;
    return seq.lastFrameTracked;
  }

  ORBextractor& leftExtractor() { return left; }
  ORBextractor& rightExtractor() { return right; }

 private:
  ORBextractor left, right;
  ORBmatcher matcher;
  OdoCamera cam;
  OdoSequence seq;

 public:
  State& state;
  std::vector<std::vector<float>>& trajectory;
  int &lastMatches, &lastMapMatches, &lastLocalInliers;
};

// ---- many independent sequences in lockstep ----
class StereoOdometryBatch {
 public:
  StereoOdometryBatch(int nSequences, float fx, float fy, float cx, float cy, float bf, int width, int height, float thDepth = 35.f,
                      int nFeatures = 2000, float scale = 1.2f, int nLevels = 8, int iniTh = 20, int minTh = 5, int device = 0)
      : nseq(nSequences), w(width), h(height), device_(device), matcher(0.9f, true, device) {
    ps_orb_config cfg{nFeatures, scale, nLevels, iniTh, minTh, 2 * nSequences, device};
    if (ps_orb_create(&cfg, &orb) != PS_OK) throw std::runtime_error(std::string("ps_orb_create: ") + ps_last_error());
    // an optimiser handle of its own (not the process-wide one of the static Optimizer functions): several batches may run
    // on different threads
    if (ps_optimizer_create(device, &opt) != PS_OK) { ps_orb_destroy(orb); throw std::runtime_error(std::string("ps_optimizer_create: ") + ps_last_error()); }
    std::vector<float> sfv(nLevels), isf(nLevels), s2(nLevels), is2(nLevels);
    ps_orb_get_tables(orb, sfv.data(), isf.data(), s2.data(), is2.data(), nullptr);
    cam.reset(new OdoCamera(fx, fy, cx, cy, bf, width, height, thDepth, sfv, is2));
    for (int k = 0; k < nSequences; k++) seqs.emplace_back(new OdoSequence(cam.get()));
    cap = nFeatures + 4 * nLevels + 64;
    nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)std::min(nSequences, 16));
    for (int t = 1; t < nthreads; t++) workers.emplace_back([this, t]() { workerLoop(t); });
  }
  ~StereoOdometryBatch() {
    {
      std::lock_guard<std::mutex> lk(mtx);
      quit = true;
    }
    cvStart.notify_all();
    for (std::thread& th : workers) th.join();
    ps_optimizer_destroy(opt);
    ps_orb_destroy(orb);
  }
  StereoOdometryBatch(const StereoOdometryBatch&) = delete;

  int size() const { return nseq; }
  OdoSequence& sequence(int k) { return *seqs[k]; }
  // wall-clock seconds spent per part of TrackAll, accumulated over calls
  double tExtract = 0, tHost = 0, tSearch = 0, tPose = 0;
  int rounds = 0;

  // One stereo frame of every sequence: left[k] / right[k] are w x h 8-bit images, rows `stride` bytes apart (buffers from
  // ps_pinned_alloc make the upload asynchronous).  Returns the number of sequences whose frame has a pose.
  // nextLeft / nextRight (optional): the FOLLOWING step's images.  Their upload, extraction and stereo matching are queued as
  // soon as this step's frames have been read back, so they overlap with this step's search / pose rounds; the next call must
  // then pass exactly those images as left / right.
  int TrackAll(const std::vector<const uint8_t*>& left, const std::vector<const uint8_t*>& right, int stride,
               const std::vector<const uint8_t*>* nextLeft = nullptr, const std::vector<const uint8_t*>* nextRight = nullptr) {
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    if ((int)left.size() != nseq || (int)right.size() != nseq) throw std::runtime_error("TrackAll: one image pair per sequence");
    const auto t0 = clk::now();
    // Frame::Frame for all sequences: ExtractORB x 2 and ComputeStereoMatches as one batch, one read-back
    if (submitted.empty()) {
      submit(left, right, stride);
    } else if (submitted[0] != left[0] || submitted[1] != right[0]) {
      throw std::runtime_error("TrackAll: the images differ from the ones announced as next in the previous call");
    }
    submitted.clear();
    std::vector<std::unique_ptr<OdoFrame>> frames(nseq);
    std::vector<ps_stereo_frame> sf(nseq);
    parallelFor([&](int k) {
      frames[k].reset(new OdoFrame);
      OdoFrame& F = *frames[k];
      F.mvKeys.resize(cap); F.mDescriptors.create(cap, 32, 0); F.mvuRight.assign(cap, -1.f); F.mvDepth.assign(cap, -1.f);
      sf[k] = ps_stereo_frame{(ps_keypoint*)F.mvKeys.data(), F.mDescriptors.data, F.mvuRight.data(), F.mvDepth.data(), cap, 0, 0, 0};
    });
    check(ps_orb_stereo_fetch_frames(orb, sf.data(), nseq));
    if (nextLeft && nextRight) {
      if ((int)nextLeft->size() != nseq || (int)nextRight->size() != nseq) throw std::runtime_error("TrackAll: one next image pair per sequence");
      submit(*nextLeft, *nextRight, stride);
      submitted = {(*nextLeft)[0], (*nextRight)[0]};
    }
    const auto t1 = clk::now();
    tExtract += secs(t0, t1);
    std::vector<OdoSequence::Request> rq(nseq, OdoSequence::NONE);
    parallelFor([&](int k) {
      OdoFrame& F = *frames[k];
      const int n = sf[k].n;
      F.mvKeys.resize(n); F.mvuRight.resize(n); F.mvDepth.resize(n); F.mDescriptors.rows = n;
      rq[k] = seqs[k]->begin(std::move(frames[k]));
    });
    auto t2 = clk::now();
    tHost += secs(t1, t2);
    std::vector<ps_proj_problem> searches;
    std::vector<ps_pose_problem> poses;
    std::vector<int> owner;
    for (;;) {
      // every pending SearchByProjection of the round in one call, then every pending PoseOptimization in one call
      searches.clear(); owner.clear();
      for (int k = 0; k < nseq; k++) if (rq[k] == OdoSequence::SEARCH) { searches.push_back(seqs[k]->proj); owner.push_back(k); }
      if (!searches.empty()) {
        matcher.SearchByProjectionBatch(searches.data(), (int)searches.size());
        for (size_t i = 0; i < owner.size(); i++) seqs[owner[i]]->proj.nmatches = searches[i].nmatches;
      }
      auto t3 = clk::now();
      tSearch += secs(t2, t3);
      poses.clear(); owner.clear();
      for (int k = 0; k < nseq; k++) if (rq[k] == OdoSequence::POSE) { poses.push_back(seqs[k]->posep); owner.push_back(k); }
      if (!poses.empty()) {
        check(ps_pose_optimize_batch(opt, poses.data(), (int)poses.size()));
        for (size_t i = 0; i < owner.size(); i++) seqs[owner[i]]->posep = poses[i];
      }
      auto t4 = clk::now();
      tPose += secs(t3, t4);
      if (searches.empty() && poses.empty()) break;
      rounds++;
      parallelFor([&](int k) { if (rq[k] != OdoSequence::NONE) rq[k] = seqs[k]->advance(); });
      t2 = clk::now();
      tHost += secs(t4, t2);
    }
    int tracked = 0;
    for (int k = 0; k < nseq; k++) tracked += seqs[k]->lastFrameTracked ? 1 : 0;
    return tracked;
  }

 private:
  int nseq, w, h, device_, cap = 0, nthreads = 1;
  ps_orb* orb = nullptr;
  ps_optimizer* opt = nullptr;
  ORBmatcher matcher;
  std::unique_ptr<OdoCamera> cam;
  std::vector<std::unique_ptr<OdoSequence>> seqs;

  static void check(int rc) { if (rc != PS_OK) throw std::runtime_error(ps_last_error()); }
  std::vector<const uint8_t*> submitted;   // first left / right image of a step whose extraction is already queued
  void submit(const std::vector<const uint8_t*>& left, const std::vector<const uint8_t*>& right, int stride) {
    std::vector<const uint8_t*> imgs(2 * (size_t)nseq);
    for (int k = 0; k < nseq; k++) { imgs[2 * k] = left[k]; imgs[2 * k + 1] = right[k]; }
    check(ps_orb_extract_batch(orb, imgs.data(), 2 * nseq, w, h, stride));
    check(ps_orb_stereo_match_batch(orb, nseq, cam->mb, cam->mbf));
  }
  // the sequences' host stages run on a persistent pool (the calling thread is worker 0)
  std::vector<std::thread> workers;
  std::mutex mtx;
  std::condition_variable cvStart, cvDone;
  std::function<void(int)> job;
  long generation = 0;
  int pending = 0;
  bool quit = false;

  void workerLoop(int t) {
    long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mtx);
        cvStart.wait(lk, [&]() { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
      }
      for (int k = t; k < nseq; k += nthreads) job(k);
      {
        std::lock_guard<std::mutex> lk(mtx);
        if (--pending == 0) cvDone.notify_one();
      }
    }
  }
  template <typename Fn> void parallelFor(Fn fn) {
    if (nthreads <= 1) { for (int k = 0; k < nseq; k++) fn(k); return; }
    {
      std::lock_guard<std::mutex> lk(mtx);
      job = fn;
      pending = nthreads - 1;
      generation++;
    }
    cvStart.notify_all();
    for (int k = 0; k < nseq; k += nthreads) fn(k);
    std::unique_lock<std::mutex> lk(mtx);
    cvDone.wait(lk, [&]() { return pending == 0; });
  }
};

// ---- many independent sequences in lockstep, the whole chain resident on the device ----
// Same frames, same results as StereoOdometryBatch, but a step is one ps_tracker_step call that only queues work: the host
// side of Tracking::Track (grid, UpdateLastFrame, match application, isInFrustum, motion model) runs in the glue kernels of
// the library between the hot-path kernels, nothing is packed or copied per call, and results are read when asked for.
class StereoOdometryDevice {
 public:
  // maxObjects > 0 (at most 16 detections per frame; maxMapObjects 0 = 8 MapObjects per sequence, at most 64): the handle also carries the object half of Tracking::Track in SLOT.MODE 4 (TrackAllSlotDevice)
  StereoOdometryDevice(int nSequences, float fx, float fy, float cx, float cy, float bf, int width, int height, int maxFrames, float thDepth = 35.f,
                       int nFeatures = 2000, float scale = 1.2f, int nLevels = 8, int iniTh = 20, int minTh = 5, int device = 0, int maxObjects = 0, int maxMapObjects = 0)
      : nseq(nSequences), nobj(maxObjects) {
    ps_tracker_config cfg{nSequences, width, height, fx, fy, cx, cy, bf, thDepth, nFeatures, scale, nLevels, iniTh, minTh, maxFrames, device, maxObjects, maxMapObjects};
    if (ps_tracker_create(&cfg, &trk) != PS_OK) throw std::runtime_error(std::string("ps_tracker_create: ") + ps_last_error());
  }
  ~StereoOdometryDevice() { ps_tracker_destroy(trk); }
  StereoOdometryDevice(const StereoOdometryDevice&) = delete;

  int size() const { return nseq; }
  ps_tracker* handle() { return trk; }
  // queues one stereo frame of every sequence (host images; buffers from ps_pinned_alloc make the upload asynchronous)
  void TrackAll(const std::vector<const uint8_t*>& left, const std::vector<const uint8_t*>& right, int stride) {
    if ((int)left.size() != nseq || (int)right.size() != nseq) throw std::runtime_error("TrackAll: one image pair per sequence");
    check(ps_tracker_step(trk, left.data(), right.data(), stride));
  }
  // ... or images that already are in device memory: sequence k's left image at d_imgs + 2k * pitch, right one pitch further
  void TrackAllDevice(const uint8_t* d_imgs, int stride, size_t pitch) { check(ps_tracker_step_device(trk, d_imgs, stride, pitch)); }
  // SLOT.MODE 4: the camera chain on the background keypoints and the object chain behind it.  d_masks: the
  // instance-id images of Segmentation/ (one per sequence, rows mask_stride apart, images mask_pitch apart), d_dets: [size()][maxObjects]
  // detections of the frame (id < 0: empty slot), everything in device memory
  void TrackAllSlotDevice(const uint8_t* d_imgs, int stride, size_t pitch, const uint8_t* d_masks, int mask_stride, size_t mask_pitch, const ps_detection* d_dets) {
    check(ps_tracker_step_slot_device(trk, d_imgs, stride, pitch, d_masks, mask_stride, mask_pitch, d_dets));
  }
  // one `Car` row of ObjectTracking.txt as Tracking::ReadKittiObjectInfo (src/Tracking.cc:485-640) and DetectionObject's constructor
  // (src/DetectionObject.cc:22-73) turn it into a detection: mrectBBox = cv::Rect of the truncated doubles, mScale = (length, height,
  // width), mTruthPosInCameraFrame = fromMinimalVector(X, Y - height / 2, Z, 0, ry, 0)
  static ps_detection MakeDetection(int trackId, double x1, double y1, double x2, double y2, double h, double w, double l, double X, double Y, double Z, double ry) {
    ps_detection d{};
    d.id = trackId;
    d.bbox[0] = (int)x1; d.bbox[1] = (int)y1; d.bbox[2] = (int)(x2 - x1); d.bbox[3] = (int)(y2 - y1);
    d.scale[0] = l; d.scale[1] = h; d.scale[2] = w;
    // zyx Euler (roll 0, pitch ry, yaw 0) -> quaternion (matrix_utils.cc:18-31), w >= 0, unit norm (SE3Quat::normalizeRotation)
    double q[4] = {0.0, std::sin(ry * 0.5), 0.0, std::cos(ry * 0.5)};
    if (q[3] < 0) { q[1] = -q[1]; q[3] = -q[3]; }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    d.pose7[0] = X; d.pose7[1] = Y - h / 2; d.pose7[2] = Z;
    for (int i = 0; i < 4; i++) d.pose7[3 + i] = q[i] / n;
    return d;
  }
  // blocks; objects[(step * size() + k) * maxObjects + slot]
  void FetchObjects(std::vector<ps_object_stat>& objects) {
    const int n = Steps();
    objects.assign((size_t)n * nseq * (nobj > 0 ? nobj : 0), ps_object_stat{});
    if (nobj > 0 && n > 0) check(ps_tracker_fetch_objects(trk, 0, n, objects.data()));
  }
  void Sync() { check(ps_tracker_sync(trk)); }
  int Steps() const { int n = 0; ps_tracker_steps(trk, &n); return n; }
  // blocks; trajectories[k][step] = Tcw (16 floats), empty for a frame without a pose; stats[step * size() + k]
  void Fetch(std::vector<std::vector<std::vector<float>>>& trajectories, std::vector<ps_track_stat>* stats = nullptr) {
    const int n = Steps();
    std::vector<float> tcw((size_t)n * nseq * 16);
    std::vector<ps_track_stat> st((size_t)n * nseq);
    check(ps_tracker_fetch(trk, 0, n, tcw.data(), st.data()));
    trajectories.assign(nseq, {});
    for (int k = 0; k < nseq; k++)
      for (int i = 0; i < n; i++) {
        const size_t o = (size_t)i * nseq + k;
        trajectories[k].push_back(st[o].tracked ? std::vector<float>(tcw.begin() + o * 16, tcw.begin() + o * 16 + 16) : std::vector<float>());
      }
    if (stats) *stats = st;
  }

 private:
  int nseq, nobj = 0;
  ps_tracker* trk = nullptr;
  static void check(int rc) { if (rc != PS_OK) throw std::runtime_error(ps_last_error()); }
};

}  // namespace ORB_SLAM2
