// ORB_SLAM2::ORBmatcher hot members (/root/reference/include/ORBmatcher.h:47-118) on the C-ABI.  The reference's
// methods take Frame& and write MapPoint* into it; Frame / MapPoint are outside the hot path and are not rebuilt.  Two
// layers: (i) the reference's own signatures as templates over the caller's Frame / MapPoint types (the marshalling of
// INTEGRATION.md section 2 and the pointer write-back, tested on tests/cpp/frame_view.h), (ii) underneath, methods that
// take the ps_*_problem structs directly for callers that already keep their data as arrays.
#pragma once
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "slotcv.h"

namespace ORB_SLAM2 {

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, TH_HIGH_FORDYNAMIC = 130, RADIUS_FORDYNAMIC = 5, HISTO_LENGTH = 30;

  ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    if (ps_matcher_create(device, &h_) != PS_OK) throw std::runtime_error(std::string("ps_matcher_create: ") + ps_last_error());
  }
  ~ORBmatcher() { ps_matcher_destroy(h_); }
  ORBmatcher(const ORBmatcher&) = delete;

  // Computes the Hamming distance between two ORB descriptors (ORBmatcher.cc:2704-2720).  A single pair stays on the
  // host (same SWAR arithmetic); the bulk form is DescriptorDistanceMatrix.
  static int DescriptorDistance(const pscv::Mat& a, const pscv::Mat& b) {
    const int32_t* pa = a.ptr<int32_t>();
    const int32_t* pb = b.ptr<int32_t>();
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
      unsigned int v = *pa ^ *pb;
      v = v - ((v >> 1) & 0x55555555);
      v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
      dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
  }
  void DescriptorDistanceMatrix(const pscv::Mat& q, const pscv::Mat& t, std::vector<uint16_t>& out) {
    out.resize((size_t)q.rows * t.rows);
    if (ps_hamming_matrix(h_, q.data, q.rows, t.data, t.rows, out.data()) != PS_OK) throw std::runtime_error(ps_last_error());
  }

  // SearchByBruceMatching(LastFrame, CurrentFrame, nLastOrder, nCurrenOrder, matches) for a batch of objects:
  // fills ps_bf_problem::query_of_train / nmatches (see include/pointslot_hip.h).  Returns the summed match count.
  int SearchByBruceMatching(std::vector<ps_bf_problem>& objects) {
    if (objects.empty()) return 0;
    if (ps_match_bruteforce(h_, objects.data(), (int)objects.size(), mfNNratio, mbCheckOrientation ? 1 : 0) != PS_OK)
      throw std::runtime_error(ps_last_error());
    int n = 0;
    for (const ps_bf_problem& o : objects) n += o.nmatches;
    return n;
  }

  // The three SearchByProjection overloads; the caller fills ps_proj_problem from its Frame.
  // th_dist / ratio_test / check_orientation are set here from the overload being emulated.
  static void ConfigureProjectionFrame(ps_proj_problem& p, float nnratio, bool checkOri) {   // (Frame& cur, const Frame& last, th, bMono)
    p.frame_mode = 1; p.th_dist = TH_HIGH; p.ratio_test = 0; p.nn_ratio = nnratio; p.check_orientation = checkOri ? 1 : 0; p.use_bbox = 0;
  }
  static void ConfigureProjectionPoints(ps_proj_problem& p, float nnratio) {                 // (Frame& F, const vector<MapPoint*>&, th)
    p.frame_mode = 0; p.th_dist = TH_HIGH; p.ratio_test = 1; p.nn_ratio = nnratio; p.check_orientation = 0; p.use_bbox = 0;
  }
  static void ConfigureProjectionObject(ps_proj_problem& p, float nnratio) {                 // (Frame& F, nOrder, const vector<MapObjectPoint*>&, th)
    p.frame_mode = 0; p.th_dist = TH_HIGH_FORDYNAMIC; p.ratio_test = 1; p.nn_ratio = nnratio; p.check_orientation = 0; p.use_bbox = 1;
  }
  int SearchByProjectionFrame(ps_proj_problem& p) { ConfigureProjectionFrame(p, mfNNratio, mbCheckOrientation); return run(p); }
  int SearchByProjectionPoints(ps_proj_problem& p) { ConfigureProjectionPoints(p, mfNNratio); return run(p); }
  int SearchByProjectionObject(ps_proj_problem& p) { ConfigureProjectionObject(p, mfNNratio); return run(p); }
  // already-configured problems (possibly of different overloads / matcher settings) in one call; returns the summed match count
  int SearchByProjectionBatch(ps_proj_problem* p, int n) {
    if (n <= 0) return 0;
    if (ps_search_by_projection(h_, p, n) != PS_OK) throw std::runtime_error(ps_last_error());
    int total = 0;
    for (int i = 0; i < n; i++) total += p[i].nmatches;
    return total;
  }
  ps_matcher* handle() { return h_; }   // (kernel time of the last call: ps_matcher_last_kernel_ms)
  static float RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }   // ORBmatcher.cc:252-258

  // ------------------------------------------------------------------------------------------------------------------
  // The reference's own signatures (/root/reference/include/ORBmatcher.h:47-103) as templates over the caller's Frame /
  // MapPoint / MapObjectPoint types: any type with the reference's member names compiles (Frame.h: N, mvKeys, mvKeysUn,
  // mvuRight, mDescriptors, mvpMapPoints, mvbOutlier, mGrid, mnMinX.., mfGridElementWidthInv.., mTcw, fx.., mbf, mb,
  // mvScaleFactors, and the per-object mvObjKeys / mvObjKeysUn / mvuObjKeysRight / mvObjPointsDescriptors /
  // mvpMapObjectPoints / mvbObjKeysOutlier / mvObjKeysGrid / isInBBox).  Each does what INTEGRATION.md section 2 lists:
  // gather the fields the reference reads into a ps_proj_problem / ps_bf_problem, one C-ABI call, pointer write-back.
  // ------------------------------------------------------------------------------------------------------------------

  // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)            ORBmatcher.cc:68-155
  template <class FrameT, class MapPointT>
  int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th) {
    TrainSide T;
    fillTrain(T, F.mvKeysUn, F.mvuRight, F.mDescriptors, F.mGrid, F);
    for (size_t j = 0; j < F.mvpMapPoints.size(); j++) T.occupied[j] = (F.mvpMapPoints[j] && F.mvpMapPoints[j]->Observations() > 0) ? 1 : 0;
    return searchPoints(T, F, vpMapPoints, th, false, [&](int j, MapPointT* p) { F.mvpMapPoints[j] = p; });
  }

  // int SearchByProjection(Frame &F, const size_t &nOrder, const vector<MapObjectPoint*> &vpMapPoints, const float th)   :157-248
  template <class FrameT, class MapObjectPointT>
  int SearchByProjection(FrameT& F, const std::size_t& nOrder, const std::vector<MapObjectPointT*>& vpMapPoints, const float th) {
    if (F.mvDetectionObjects[nOrder] == NULL) throw std::runtime_error("SearchByProjection: no detection at nOrder");   // assert(0) in the reference
    TrainSide T;
    fillTrain(T, F.mvObjKeysUn[nOrder], F.mvuObjKeysRight[nOrder], F.mvObjPointsDescriptors[nOrder], F.mvObjKeysGrid[nOrder], F);
    T.in_bbox.assign(T.n, 0);
    for (int j = 0; j < T.n; j++) {
      T.in_bbox[j] = F.isInBBox(nOrder, F.mvObjKeysUn[nOrder][j].pt.x, F.mvObjKeysUn[nOrder][j].pt.y) ? 1 : 0;
      T.occupied[j] = (F.mvpMapObjectPoints[nOrder][j] && F.mvpMapObjectPoints[nOrder][j]->Observations() > 0) ? 1 : 0;
    }
    return searchPoints(T, F, vpMapPoints, th, true, [&](int j, MapObjectPointT* p) { F.mvpMapObjectPoints[nOrder][j] = p; });
  }

  // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)               :1613-1756
  template <class FrameT>
  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
    TrainSide T;
    fillTrain(T, CurrentFrame.mvKeysUn, CurrentFrame.mvuRight, CurrentFrame.mDescriptors, CurrentFrame.mGrid, CurrentFrame);
    for (size_t j = 0; j < CurrentFrame.mvpMapPoints.size(); j++)
      T.occupied[j] = (CurrentFrame.mvpMapPoints[j] && CurrentFrame.mvpMapPoints[j]->Observations() > 0) ? 1 : 0;
    const int nq = LastFrame.N;
    std::vector<uint8_t> qvalid(nq, 0), qobs(nq, 0), qdesc((size_t)nq * 32, 0);
    std::vector<float> qxw((size_t)nq * 3, 0.f), qang(nq, 0.f);
    std::vector<int32_t> qoct(nq, 0);
    for (int i = 0; i < nq; i++) {
      auto* pMP = LastFrame.mvpMapPoints[i];
      qoct[i] = LastFrame.mvKeys[i].octave; qang[i] = LastFrame.mvKeysUn[i].angle;
      if (!pMP || LastFrame.mvbOutlier[i]) continue;
      qvalid[i] = 1; qobs[i] = pMP->Observations() > 0 ? 1 : 0;
      const pscv::Mat x3Dw = pMP->GetWorldPos();
      for (int c = 0; c < 3; c++) qxw[3 * (size_t)i + c] = x3Dw.template at<float>(c);
      const pscv::Mat d = pMP->GetDescriptor();
      std::memcpy(&qdesc[(size_t)i * 32], d.template ptr<uint8_t>(), 32);
    }
    ps_proj_problem p = ps_proj_problem{};
    T.bind(p.train);
    p.nq = nq; p.q_valid = qvalid.data(); p.q_desc = qdesc.data(); p.q_observed = qobs.data(); p.q_angle = qang.data();
    p.q_xw = qxw.data(); p.q_octave = qoct.data(); p.mono = bMono ? 1 : 0;
    copyPose(CurrentFrame.mTcw, p.tcw); copyPose(LastFrame.mTcw, p.tlw);
    p.fx = CurrentFrame.fx; p.fy = CurrentFrame.fy; p.cx = CurrentFrame.cx; p.cy = CurrentFrame.cy; p.mbf = CurrentFrame.mbf; p.mb = CurrentFrame.mb;
    p.bounds[0] = CurrentFrame.mnMinX; p.bounds[1] = CurrentFrame.mnMaxX; p.bounds[2] = CurrentFrame.mnMinY; p.bounds[3] = CurrentFrame.mnMaxY;
    for (int l = 0; l < 8; l++) p.scale_factors[l] = l < (int)CurrentFrame.mvScaleFactors.size() ? CurrentFrame.mvScaleFactors[l] : 1.f;
    p.th = th;
    std::vector<int32_t> match((size_t)std::max(T.n, 1), -1);
    p.match_of_train = match.data();
    const int n = SearchByProjectionFrame(p);
    for (int j = 0; j < T.n; j++) {
      if (match[j] >= 0) CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[match[j]];
      else if (match[j] == -2) CurrentFrame.mvpMapPoints[j] = NULL;   // assigned, then reset by the rotation check (:1742-1750)
    }
    return n;
  }

  // int SearchByBruceMatching(const Frame& LastFrame, const Frame& CurrentFrame, const int &nLastOrder, const int &nCurrenOrder,
  //                           vector<MapObjectPoint *> &vpMapObjectPointMatches)                                          :2043-2155
  template <class FrameT, class MapObjectPointT>
  int SearchByBruceMatching(const FrameT& LastFrame, const FrameT& CurrentFrame, const int& nLastOrder, const int& nCurrenOrder,
                            std::vector<MapObjectPointT*>& vpMapObjectPointMatches) {
    const std::vector<MapObjectPointT*>& vpMapLast = LastFrame.mvpMapObjectPoints[nLastOrder];
    const int nq = (int)vpMapLast.size(), nt = (int)CurrentFrame.mvObjKeysUn[nCurrenOrder].size();
    vpMapObjectPointMatches.assign(CurrentFrame.mvpMapObjectPoints[nCurrenOrder].size(), static_cast<MapObjectPointT*>(NULL));
    std::vector<uint8_t> qvalid(std::max(nq, 1), 0);
    std::vector<float> qang(std::max(nq, 1), 0.f), tang(std::max(nt, 1), 0.f);
    for (int i = 0; i < nq; i++) {
      MapObjectPointT* pMP = vpMapLast[i];
      qvalid[i] = (pMP && !pMP->isBad() && !LastFrame.mvbObjKeysOutlier[nLastOrder][i]) ? 1 : 0;
      qang[i] = LastFrame.mvObjKeysUn[nLastOrder][i].angle;
    }
    for (int j = 0; j < nt; j++) tang[j] = CurrentFrame.mvObjKeys[nCurrenOrder][j].angle;
    std::vector<uint8_t> qd = rows32(LastFrame.mvObjPointsDescriptors[nLastOrder], nq), td = rows32(CurrentFrame.mvObjPointsDescriptors[nCurrenOrder], nt);
    std::vector<int32_t> qot(std::max(nt, 1), -1);
    std::vector<ps_bf_problem> pr(1);
    pr[0] = ps_bf_problem{qd.data(), qang.data(), qvalid.data(), nq, td.data(), tang.data(), nt, qot.data(), 0};
    if (nq == 0 || nt == 0) return 0;
    const int n = SearchByBruceMatching(pr);
    for (int j = 0; j < nt && j < (int)vpMapObjectPointMatches.size(); j++)
      if (qot[j] >= 0) vpMapObjectPointMatches[j] = vpMapLast[qot[j]];
    return n;
  }

 protected:
  // the "train" half of a windowed search: mvKeysUn-like keys, mvuRight, descriptors and the 64 x 48 grid of the frame (or of
  // one of its objects) as the arrays of ps_proj_train
  struct TrainSide {
    int n = 0;
    std::vector<float> x, y, angle, ur;
    std::vector<int32_t> octave, cell_off, cell_idx;
    std::vector<uint8_t> desc, occupied, in_bbox;
    float min_x = 0, min_y = 0, gw = 0, gh = 0;
    void bind(ps_proj_train& t) const {
      t.n = n; t.x = x.data(); t.y = y.data(); t.octave = octave.data(); t.angle = angle.data(); t.u_right = ur.data(); t.desc = desc.data();
      t.occupied = occupied.data(); t.in_bbox = in_bbox.empty() ? nullptr : in_bbox.data(); t.cell_off = cell_off.data(); t.cell_idx = cell_idx.data();
      t.min_x = min_x; t.min_y = min_y; t.grid_w_inv = gw; t.grid_h_inv = gh;
    }
  };
  static std::vector<uint8_t> rows32(const pscv::Mat& m, int n) {
    std::vector<uint8_t> out((size_t)std::max(n, 1) * 32, 0);
    for (int i = 0; i < n && i < m.rows; i++) std::memcpy(&out[(size_t)i * 32], m.ptr<uint8_t>(i), 32);
    return out;
  }
  template <class MatT> static void copyPose(const MatT& Tcw, float* out16) {
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) out16[4 * r + c] = Tcw.template at<float>(r, c);
  }
  template <class KeysT, class GridT, class FrameT>
  static void fillTrain(TrainSide& T, const KeysT& keysUn, const std::vector<float>& uRight, const pscv::Mat& descriptors, const GridT& grid, const FrameT& F) {
    const int n = (int)keysUn.size();
    T.n = n;
    T.x.resize(std::max(n, 1)); T.y.resize(std::max(n, 1)); T.angle.resize(std::max(n, 1)); T.ur.resize(std::max(n, 1)); T.octave.resize(std::max(n, 1));
    T.occupied.assign(std::max(n, 1), 0);
    for (int j = 0; j < n; j++) {
      T.x[j] = keysUn[j].pt.x; T.y[j] = keysUn[j].pt.y; T.angle[j] = keysUn[j].angle; T.octave[j] = keysUn[j].octave; T.ur[j] = uRight[j];
    }
    T.desc = rows32(descriptors, n);
    const int GC = 64, GR = 48;   // FRAME_GRID_COLS / FRAME_GRID_ROWS (Frame.h:40-41)
    T.cell_off.assign(GC * GR + 1, 0);
    T.cell_idx.clear();
    for (int ix = 0; ix < GC; ix++)
      for (int iy = 0; iy < GR; iy++) {
        T.cell_off[ix * GR + iy] = (int32_t)T.cell_idx.size();
        for (std::size_t k : grid[ix][iy]) T.cell_idx.push_back((int32_t)k);
      }
    T.cell_off[GC * GR] = (int32_t)T.cell_idx.size();
    T.cell_idx.resize(std::max<size_t>(T.cell_idx.size(), (size_t)std::max(n, 1)), 0);
    T.min_x = F.mnMinX; T.min_y = F.mnMinY; T.gw = F.mfGridElementWidthInv; T.gh = F.mfGridElementHeightInv;
  }
  // the two map-point overloads share everything but the window (r * scale[level] and levels [l-1, l] for map points; 5 px and
  // [l-1, l+1] for object points, ORBmatcher.cc:92-93 / :182) and the threshold
  template <class FrameT, class PointT, class Assign>
  int searchPoints(TrainSide& T, const FrameT& F, const std::vector<PointT*>& pts, const float th, bool object, Assign assign) {
    const int nq = (int)pts.size();
    std::vector<uint8_t> qvalid(std::max(nq, 1), 0), qobs(std::max(nq, 1), 0), qdesc((size_t)std::max(nq, 1) * 32, 0);
    std::vector<float> qu(std::max(nq, 1), 0.f), qv(std::max(nq, 1), 0.f), qur(std::max(nq, 1), 0.f), rad(std::max(nq, 1), 0.f), rer(std::max(nq, 1), 0.f);
    std::vector<int32_t> minl(std::max(nq, 1), 0), maxl(std::max(nq, 1), 0);
    const bool bFactor = th != 1.0;
    for (int i = 0; i < nq; i++) {
      PointT* pMP = pts[i];
      if (!pMP->mbTrackInView || pMP->isBad()) continue;
      const int nPredictedLevel = pMP->mnTrackScaleLevel;
      float r = RadiusByViewingCos(pMP->mTrackViewCos);
      if (bFactor) r *= th;
      qvalid[i] = 1; qobs[i] = pMP->Observations() > 0 ? 1 : 0;
      qu[i] = pMP->mTrackProjX; qv[i] = pMP->mTrackProjY; qur[i] = pMP->mTrackProjXR;
      rer[i] = r * F.mvScaleFactors[nPredictedLevel];
      rad[i] = object ? (float)RADIUS_FORDYNAMIC : rer[i];
      minl[i] = nPredictedLevel - 1; maxl[i] = object ? nPredictedLevel + 1 : nPredictedLevel;
      const pscv::Mat d = pMP->GetDescriptor();
      std::memcpy(&qdesc[(size_t)i * 32], d.template ptr<uint8_t>(), 32);
    }
    ps_proj_problem p = ps_proj_problem{};
    T.bind(p.train);
    p.nq = nq; p.q_valid = qvalid.data(); p.q_desc = qdesc.data(); p.q_observed = qobs.data();
    p.q_u = qu.data(); p.q_v = qv.data(); p.q_ur = qur.data(); p.q_radius = rad.data(); p.q_radius_er = rer.data();
    p.q_min_level = minl.data(); p.q_max_level = maxl.data();
    for (int l = 0; l < 8; l++) p.scale_factors[l] = l < (int)F.mvScaleFactors.size() ? F.mvScaleFactors[l] : 1.f;
    p.th = th;
    std::vector<int32_t> match((size_t)std::max(T.n, 1), -1);
    p.match_of_train = match.data();
    if (nq == 0) return 0;
    const int n = object ? SearchByProjectionObject(p) : SearchByProjectionPoints(p);
    for (int j = 0; j < T.n; j++)
      if (match[j] >= 0) assign(j, pts[match[j]]);
    return n;
  }

  int run(ps_proj_problem& p) {
    if (ps_search_by_projection(h_, &p, 1) != PS_OK) throw std::runtime_error(ps_last_error());
    return p.nmatches;
  }
  ps_matcher* h_ = nullptr;
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2
