// The tracking thread's data flow around the hot path in C++ (the same slice as pointslot_amd/tracker.py, which documents
// the mapping to /root/reference/src/Tracking.cc:2840-3160,1260-1286 and src/Frame.cc:1686-1743,2505-2519):
//   Frame::Frame (ExtractORB x2 + ComputeStereoMatches), StereoInitialization, UpdateLastFrame (localisation mode),
//   TrackWithMotionModel, SearchLocalPoints / TrackLocalMap, the constant-velocity model.
// Host code only; every heavy step is one call into the C-ABI through the shim classes.  float arithmetic where the
// reference uses cv::Mat CV_32F.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
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

class StereoOdometry {
 public:
  enum State { NOT_INITIALIZED, OK, LOST };
  static const int GRID_COLS = 64, GRID_ROWS = 48;

  StereoOdometry(float fx, float fy, float cx, float cy, float bf, int width, int height, float thDepth = 35.f, int nFeatures = 2000,
                 float scale = 1.2f, int nLevels = 8, int iniTh = 20, int minTh = 5)
      : fx(fx), fy(fy), cx(cx), cy(cy), mbf(bf), w(width), h(height), left(nFeatures, scale, nLevels, iniTh, minTh),
        right(nFeatures, scale, nLevels, iniTh, minTh), matcherMM(0.9f, true), matcherLM(0.8f, true) {
    mb = mbf / fx;
    mThDepth = mbf * thDepth / fx;                                                  // Tracking.cc:402
    gwInv = (float)GRID_COLS / (float)width; ghInv = (float)GRID_ROWS / (float)height;   // Frame.cc:1636-1640, no distortion
    sf = left.GetScaleFactors(); invSigma2 = left.GetInverseScaleSigmaSquares();
    logSf = std::log(sf[1]);
    left.mbDownloadPyramid = false; right.mbDownloadPyramid = false;
  }

  State state = NOT_INITIALIZED;
  std::vector<std::vector<float>> trajectory;   // Tcw (16 floats) per frame, empty when lost / not initialised

  // Tracking::Track for one stereo frame; returns true when the frame has a pose
  bool Track(const pscv::Mat& imLeft, const pscv::Mat& imRight) {
    std::unique_ptr<OdoFrame> F(new OdoFrame);
    makeFrame(*F, imLeft, imRight);
    if (state == NOT_INITIALIZED) {
      if (initialize(*F)) { last = std::move(F); haveVelocity = false; trajectory.push_back(pose(*last)); return true; }
      trajectory.push_back({});
      return false;
    }
    if (!haveVelocity) { setIdentity(velocity); haveVelocity = true; }   // see tracker.py: no vocabulary for TrackReferenceKeyFrame
    bool ok = trackMotionModel(*F);
    if (ok && !mbVO) ok = trackLocalMap(*F);
    if (!ok) { state = LOST; haveVelocity = false; trajectory.push_back({}); return false; }
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
    return true;
  }

  int lastMatches = 0, lastMapMatches = 0, lastLocalInliers = 0;

 private:
  float fx, fy, cx, cy, mbf, mb, mThDepth, gwInv, ghInv, logSf;
  int w, h;
  ORBextractor left, right;
  ORBmatcher matcherMM, matcherLM;
  std::vector<float> sf, invSigma2;
  std::unique_ptr<OdoFrame> last;
  float velocity[16];
  bool haveVelocity = false, mbVO = false;
  // the initial keyframe's map points (the local map of this slice)
  std::vector<float> lm_xw, lm_normal, lm_maxd, lm_mind;
  std::vector<uint8_t> lm_desc;

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

  void makeFrame(OdoFrame& F, const pscv::Mat& imL, const pscv::Mat& imR) {
    std::vector<pscv::KeyPoint> keysR;
    pscv::Mat descR;
    // two threads, one extractor each, as the reference does (Frame.cc:709-710): the handles own separate streams, so the two
    // single-image pipelines overlap on the GPU
    std::thread threadLeft([&]() { left(imL, pscv::Mat(), F.mvKeys, F.mDescriptors); });
    std::thread threadRight([&]() { right(imR, pscv::Mat(), keysR, descR); });
    threadLeft.join();
    threadRight.join();
    F.N = (int)F.mvKeys.size();
    F.mvuRight.assign(F.N, -1.f); F.mvDepth.assign(F.N, -1.f);
    int n = 0;
    if (F.N > 0 && ps_orb_stereo_match_pair(left.handle(), right.handle(), mb, mbf, F.mvuRight.data(), F.mvDepth.data(), F.N, &n) != PS_OK)
      throw std::runtime_error(ps_last_error());                                         // Frame::ComputeStereoMatches
    F.x.resize(F.N); F.y.resize(F.N); F.angle.resize(F.N); F.octave.resize(F.N);
    for (int i = 0; i < F.N; i++) { F.x[i] = F.mvKeys[i].pt.x; F.y[i] = F.mvKeys[i].pt.y; F.angle[i] = F.mvKeys[i].angle; F.octave[i] = F.mvKeys[i].octave; }
    // Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:1636-1656, 2027-2037) as CSR, cell = ix * 48 + iy
    std::vector<int> cell(F.N, -1);
    F.cell_off.assign(GRID_COLS * GRID_ROWS + 1, 0);
    for (int i = 0; i < F.N; i++) {
      const int px = (int)std::round((F.x[i] - 0.f) * gwInv), py = (int)std::round((F.y[i] - 0.f) * ghInv);
      if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
      cell[i] = px * GRID_ROWS + py;
      F.cell_off[cell[i] + 1]++;
    }
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) F.cell_off[c + 1] += F.cell_off[c];
    F.cell_idx.assign(F.cell_off.back(), 0);
    std::vector<int> fill(F.cell_off.begin(), F.cell_off.end() - 1);
    for (int i = 0; i < F.N; i++) if (cell[i] >= 0) F.cell_idx[fill[cell[i]]++] = i;
    F.mp_xw.assign((size_t)F.N * 3, 0.f); F.mp_valid.assign(F.N, 0); F.mp_observed.assign(F.N, 0); F.outlier.assign(F.N, 0);
    F.mp_id.assign(F.N, -1);
  }

  // Frame::UnprojectStereo (Frame.cc:2505-2519)
  void unproject(const OdoFrame& F, int i, float* X) const {
    const float z = F.mvDepth[i];
    const float xc = (F.x[i] - cx) * z * (1.f / fx), yc = (F.y[i] - cy) * z * (1.f / fy);
    const float* T = F.tcw;
    float Ow[3];
    for (int r = 0; r < 3; r++) Ow[r] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
    for (int r = 0; r < 3; r++) X[r] = (T[r] * xc + T[4 + r] * yc + T[8 + r] * z) + Ow[r];   // Rwc * x3Dc + mOw
  }

  bool initialize(OdoFrame& F) {                                    // Tracking::StereoInitialization
    if (F.N <= 500) return false;
    setIdentity(F.tcw); F.has_pose = true;
    int id = 0;
    for (int i = 0; i < F.N; i++) {
      if (!(F.mvDepth[i] > 0)) continue;
      unproject(F, i, &F.mp_xw[3 * (size_t)i]);
      F.mp_valid[i] = 1; F.mp_observed[i] = 1; F.mp_id[i] = id++;
      // MapPoint::UpdateNormalAndDepth with one observation (MapPoint.cc:470-497): camera centre at the origin
      const float* P = &F.mp_xw[3 * (size_t)i];
      const float dist = std::sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
      const float maxd = dist * sf[F.octave[i]];
      for (int c = 0; c < 3; c++) { lm_xw.push_back(P[c]); lm_normal.push_back(P[c] / dist); }
      lm_maxd.push_back(maxd); lm_mind.push_back(maxd / sf.back());
      lm_desc.insert(lm_desc.end(), F.mDescriptors.ptr<uint8_t>(i), F.mDescriptors.ptr<uint8_t>(i) + 32);
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
      if (vDepthIdx[j].first > 2 * mThDepth && nPoints > 100) break;
    }
  }

  void fillTrain(ps_proj_train& t, const OdoFrame& F, const std::vector<uint8_t>& occupied) const {
    t.n = F.N; t.x = F.x.data(); t.y = F.y.data(); t.octave = F.octave.data(); t.angle = F.angle.data(); t.u_right = F.mvuRight.data();
    t.desc = F.mDescriptors.data; t.occupied = occupied.data(); t.in_bbox = nullptr; t.cell_off = F.cell_off.data(); t.cell_idx = F.cell_idx.data();
    t.min_x = 0.f; t.min_y = 0.f; t.grid_w_inv = gwInv; t.grid_h_inv = ghInv;
  }

  int poseOptimization(OdoFrame& F) {                               // Optimizer::PoseOptimization(&mCurrentFrame)
    std::vector<float> obs((size_t)F.N * 3), is2(F.N);
    int nvalid = 0;
    for (int i = 0; i < F.N; i++) {
      obs[3 * (size_t)i] = F.x[i]; obs[3 * (size_t)i + 1] = F.y[i]; obs[3 * (size_t)i + 2] = F.mvuRight[i];
      is2[i] = invSigma2[F.octave[i]];
      nvalid += F.mp_valid[i] ? 1 : 0;
    }
    ps_pose_problem p{};
    p.n = F.N; p.xw = F.mp_xw.data(); p.obs = obs.data(); p.inv_sigma2 = is2.data(); p.valid = F.mp_valid.data();
    p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.bf = mbf;
    std::memcpy(p.tcw, F.tcw, 64);
    p.outlier = F.outlier.data();
    const int r = Optimizer::PoseOptimization(&p);
    if (nvalid >= 15) std::memcpy(F.tcw, p.tcw, 64);               // Optimizer.cc:376-377
    return r;
  }

  bool trackMotionModel(OdoFrame& F) {                              // Tracking::TrackWithMotionModel
    OdoFrame& L = *last;
    updateLastFrame();
    mul4(velocity, L.tcw, F.tcw); F.has_pose = true;
    std::vector<uint8_t> qvalid(L.N), qobs(L.N, 1), occupied(F.N, 0);
    for (int i = 0; i < L.N; i++) qvalid[i] = (L.mp_valid[i] && !L.outlier[i]) ? 1 : 0;
    std::vector<int32_t> match(std::max(F.N, 1), -1);
    int nm = 0;
    for (float th : {7.f, 14.f}) {
      ps_proj_problem p{};
      fillTrain(p.train, F, occupied);
      p.nq = L.N; p.q_valid = qvalid.data(); p.q_desc = L.mDescriptors.data; p.q_observed = qobs.data(); p.q_angle = L.angle.data();
      p.q_xw = L.mp_xw.data(); p.q_octave = L.octave.data(); p.mono = 0;
      std::memcpy(p.tcw, F.tcw, 64); std::memcpy(p.tlw, L.tcw, 64);
      p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.mbf = mbf; p.mb = mb;
      p.bounds[0] = 0.f; p.bounds[1] = (float)w; p.bounds[2] = 0.f; p.bounds[3] = (float)h;
      for (int l = 0; l < 8; l++) p.scale_factors[l] = l < (int)sf.size() ? sf[l] : 1.f;
      p.th = th; p.match_of_train = match.data();
      nm = matcherMM.SearchByProjectionFrame(p);
      if (nm >= 20) break;
    }
    if (nm < 20) return false;
    for (int j = 0; j < F.N; j++) {
      const int i = match[j];
      F.mp_valid[j] = i >= 0;
      if (i >= 0) { std::memcpy(&F.mp_xw[3 * (size_t)j], &L.mp_xw[3 * (size_t)i], 12); F.mp_observed[j] = L.mp_observed[i]; F.mp_id[j] = L.mp_id[i]; }
    }
    poseOptimization(F);
    int nmatches = 0, nmatchesMap = 0;
    for (int i = 0; i < F.N; i++) {                                 // discard outliers (Tracking.cc:3062-3082)
      if (!F.mp_valid[i]) continue;
      if (F.outlier[i]) { F.mp_valid[i] = 0; F.outlier[i] = 0; continue; }
      nmatches++;
      if (F.mp_observed[i]) nmatchesMap++;
    }
    lastMatches = nmatches; lastMapMatches = nmatchesMap;
    mbVO = nmatchesMap < 10;
    return nmatches > 20;
  }

  bool trackLocalMap(OdoFrame& F) {                                 // SearchLocalPoints + TrackLocalMap
    const int n = (int)lm_maxd.size();
    std::vector<uint8_t> already(n, 0), qvalid(n, 0), qobs(n, 1), occupied(F.N, 0);
    for (int i = 0; i < F.N; i++) {
      if (F.mp_valid[i] && F.mp_id[i] >= 0) already[F.mp_id[i]] = 1;
      occupied[i] = (F.mp_valid[i] && F.mp_observed[i]) ? 1 : 0;
    }
    std::vector<float> qu(n, 0.f), qv(n, 0.f), qur(n, 0.f), rad(n, 0.f);
    std::vector<int32_t> minl(n, 0), maxl(n, 0);
    const float* T = F.tcw;
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
      if (u < 0 || u > (float)w || v < 0 || v > (float)h) continue;
      const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
      const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      if (dist < 0.8f * lm_mind[i] || dist > 1.2f * lm_maxd[i]) continue;
      const float viewCos = (PO[0] * lm_normal[3 * (size_t)i] + PO[1] * lm_normal[3 * (size_t)i + 1] + PO[2] * lm_normal[3 * (size_t)i + 2]) / dist;
      if (viewCos < 0.5f) continue;
      int level = (int)std::ceil(std::log(lm_maxd[i] / dist) / logSf);   // MapPoint::PredictScale
      level = level < 0 ? 0 : (level >= (int)sf.size() ? (int)sf.size() - 1 : level);
      float r = ORBmatcher::RadiusByViewingCos(viewCos);            // th = 1: no factor (ORBmatcher.cc:74-84)
      qvalid[i] = 1; qu[i] = u; qv[i] = v; qur[i] = u - mbf * invz; rad[i] = r * sf[level]; minl[i] = level - 1; maxl[i] = level;
      nto++;
    }
    if (nto > 0) {
      std::vector<int32_t> match(std::max(F.N, 1), -1);
      ps_proj_problem p{};
      fillTrain(p.train, F, occupied);
      p.nq = n; p.q_valid = qvalid.data(); p.q_desc = lm_desc.data(); p.q_observed = qobs.data();
      p.q_u = qu.data(); p.q_v = qv.data(); p.q_ur = qur.data(); p.q_radius = rad.data(); p.q_radius_er = rad.data();
      p.q_min_level = minl.data(); p.q_max_level = maxl.data();
      for (int l = 0; l < 8; l++) p.scale_factors[l] = l < (int)sf.size() ? sf[l] : 1.f;
      p.th = 1.f; p.match_of_train = match.data();
      matcherLM.SearchByProjectionPoints(p);
      for (int j = 0; j < F.N; j++) {
        const int i = match[j];
        if (i < 0) continue;
        F.mp_valid[j] = 1; std::memcpy(&F.mp_xw[3 * (size_t)j], &lm_xw[3 * (size_t)i], 12); F.mp_observed[j] = 1; F.mp_id[j] = i;
      }
    }
    poseOptimization(F);
    int inl = 0;
    for (int i = 0; i < F.N; i++) {
      if (!F.mp_valid[i]) continue;
      if (F.outlier[i]) F.mp_valid[i] = 0;                          // stereo: outliers lose their map point (Tracking.cc:3141-3142)
      else inl++;
    }
    lastLocalInliers = inl;
    return inl >= 30;
  }
};

}  // namespace ORB_SLAM2
