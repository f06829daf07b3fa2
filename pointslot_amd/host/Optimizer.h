// ORB_SLAM2::Optimizer hot static members (/root/reference/include/Optimizer.h:51-61) on the C-ABI.  The reference's
// functions take Frame* / ObjectKeyFrame* and mutate them.  Two layers: (i) the reference's own signatures as templates
// over the caller's types (graph collection, marshalling and write-back as INTEGRATION.md section 3 describes; tested on
// tests/cpp/frame_view.h), (ii) underneath, overloads that take the ps_pose_problem / ps_cfse3_problem / ps_ba_problem
// arrays directly (field mapping in include/pointslot_hip.h) and return what the reference returns.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "slotcv.h"

namespace ORB_SLAM2 {

class Optimizer {
 public:
  // one process-wide handle per device, created on first use (the reference's functions are static)
  static ps_optimizer* handle(int device = 0) {
    static ps_optimizer* h[16] = {};
    if (device < 0 || device >= 16) throw std::runtime_error("bad device");
    if (!h[device] && ps_optimizer_create(device, &h[device]) != PS_OK)
      throw std::runtime_error(std::string("ps_optimizer_create: ") + ps_last_error());
    return h[device];
  }
  // int Optimizer::PoseOptimization(Frame*): returns nInitialCorrespondences - nBad (0 when < 15 correspondences);
  // p.tcw / p.outlier are updated like pFrame->SetPose / mvbOutlier.
  static int PoseOptimization(ps_pose_problem* p, int device = 0) {
    if (ps_pose_optimize_batch(handle(device), p, 1) != PS_OK) throw std::runtime_error(ps_last_error());
    return p->result;
  }
  static void PoseOptimizationBatch(std::vector<ps_pose_problem>& frames, int device = 0) {
    if (!frames.empty() && ps_pose_optimize_batch(handle(device), frames.data(), (int)frames.size()) != PS_OK)
      throw std::runtime_error(ps_last_error());
  }
  // int Optimizer::CFSE3ObjStateOptimization(Frame*, vnNeedToBeOptimized, verbose): returns true/false as int
  static int CFSE3ObjStateOptimization(ps_cfse3_problem* p, const bool& /*bVerbose*/ = false, int device = 0) {
    if (ps_cfse3_optimize_batch(handle(device), p, 1) != PS_OK) throw std::runtime_error(ps_last_error());
    return p->result;
  }
  // void Optimizer::ObjectLocalBundleAdjustment(ObjectKeyFrame*, verbose) on the collected graph(s)
  static void ObjectLocalBundleAdjustment(ps_ba_problem* p, const bool& /*bVerbose*/ = false, int device = 0) {
    if (ps_object_ba_batch(handle(device), p, 1) != PS_OK) throw std::runtime_error(ps_last_error());
  }
  static void ObjectLocalBundleAdjustmentBatch(std::vector<ps_ba_problem>& objs, int device = 0) {
    if (!objs.empty() && ps_object_ba_batch(handle(device), objs.data(), (int)objs.size()) != PS_OK)
      throw std::runtime_error(ps_last_error());
  }

  // ------------------------------------------------------------------------------------------------------------------
  // The reference's own signatures (/root/reference/include/Optimizer.h:51-61) as templates over the caller's Frame /
  // ObjectKeyFrame types (any type with the reference's member names): the marshalling of INTEGRATION.md section 3, one
  // C-ABI call, and the write-back the reference does at the end of each function.
  // ------------------------------------------------------------------------------------------------------------------

  // int Optimizer::PoseOptimization(Frame *pFrame)                                                  Optimizer.cc:249-477
  template <class FrameT>
  static int PoseOptimization(FrameT* pFrame) {
    const int N = pFrame->N;
    std::vector<float> xw((size_t)std::max(N, 1) * 3, 0.f), obs((size_t)std::max(N, 1) * 3, 0.f), is2(std::max(N, 1), 0.f);
    std::vector<uint8_t> valid(std::max(N, 1), 0), outl(std::max(N, 1), 0);
    for (int i = 0; i < N; i++) {
      auto* pMP = pFrame->mvpMapPoints[i];
      outl[i] = pFrame->mvbOutlier[i] ? 1 : 0;
      obs[3 * (size_t)i] = pFrame->mvKeysUn[i].pt.x; obs[3 * (size_t)i + 1] = pFrame->mvKeysUn[i].pt.y; obs[3 * (size_t)i + 2] = pFrame->mvuRight[i];
      is2[i] = pFrame->mvInvLevelSigma2[pFrame->mvKeysUn[i].octave];
      if (!pMP) continue;
      valid[i] = 1;
      const pscv::Mat Xw = pMP->GetWorldPos();
      for (int c = 0; c < 3; c++) xw[3 * (size_t)i + c] = Xw.template at<float>(c);
    }
    ps_pose_problem p = ps_pose_problem{};
    p.n = N; p.xw = xw.data(); p.obs = obs.data(); p.inv_sigma2 = is2.data(); p.valid = valid.data();
    p.fx = pFrame->fx; p.fy = pFrame->fy; p.cx = pFrame->cx; p.cy = pFrame->cy; p.bf = pFrame->mbf;
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) p.tcw[4 * r + c] = pFrame->mTcw.template at<float>(r, c);
    p.outlier = outl.data();
    int nvalid = 0;
    for (int i = 0; i < N; i++) nvalid += valid[i];
    PoseOptimization(&p);
    for (int i = 0; i < N; i++)
      if (valid[i]) pFrame->mvbOutlier[i] = outl[i] != 0;
    if (nvalid < 15) return 0;                                    // the reference returns before SetPose (:376-377)
    pscv::Mat pose(4, 4, pscv::CV_32F);
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) pose.template at<float>(r, c) = p.tcw[4 * r + c];
    pFrame->SetPose(pose);                                        // :471-474
    return p.result;
  }

  // int Optimizer::CFSE3ObjStateOptimization(Frame *pFrame, const vector<size_t> &vnNeedToBeOptimized, const bool &bVerbose)   :479-753
  template <class FrameT>
  static int CFSE3ObjStateOptimization(FrameT* pFrame, const std::vector<std::size_t>& vnNeedToBeOptimized, const bool& bVerbose) {
    const int k = (int)vnNeedToBeOptimized.size();
    if (k == 0) return false;                                     // pMObjects.size()==0 (:504-505)
    std::vector<int32_t> off(k + 1, 0);
    for (int i = 0; i < k; i++) {
      const std::size_t n = vnNeedToBeOptimized[i];
      if (pFrame->mvMapObjects[n] == NULL) throw std::runtime_error("CFSE3ObjStateOptimization: no MapObject at the given order");   // assert(0)
      off[i + 1] = off[i] + (int)pFrame->mvpMapObjectPoints[n].size();
    }
    const int total = off[k];
    std::vector<float> xo((size_t)std::max(total, 1) * 3, 0.f), obs((size_t)std::max(total, 1) * 3, 0.f), is2(std::max(total, 1), 0.f);
    std::vector<uint8_t> valid(std::max(total, 1), 0), outl(std::max(total, 1), 0);
    std::vector<double> poses((size_t)k * 7, 0.0);
    for (int i = 0; i < k; i++) {
      const std::size_t n = vnNeedToBeOptimized[i];
      auto* pMO = pFrame->mvMapObjects[n];
      pMO->mmBAFrameIdAndObjVertexID.clear();                     // :510-520
      pMO->mmBAFrameIdAndObjVertexID[pFrame->mnId] = i;
      pMO->GetCFInFrameObjState(pFrame->mnId).pose.toVector(&poses[(size_t)i * 7]);
      const auto& vpMP = pFrame->mvpMapObjectPoints[n];
      for (size_t j = 0; j < vpMP.size(); j++) {
        const size_t e = (size_t)off[i] + j;
        const auto& kpUn = pFrame->mvObjKeysUn[n][j];
        outl[e] = pFrame->mvbObjKeysOutlier[n][j] ? 1 : 0;
        obs[3 * e] = kpUn.pt.x; obs[3 * e + 1] = kpUn.pt.y; obs[3 * e + 2] = pFrame->mvuObjKeysRight[n][j];
        is2[e] = pFrame->mvInvLevelSigma2[kpUn.octave];
        if (!vpMP[j]) continue;
        valid[e] = 1;
        const pscv::Mat Xo = vpMP[j]->GetInObjFramePosition();
        for (int c = 0; c < 3; c++) xo[3 * e + c] = Xo.template at<float>(c);
      }
    }
    ps_cfse3_problem p = ps_cfse3_problem{};
    p.k = k; p.off = off.data(); p.xo = xo.data(); p.obs = obs.data(); p.inv_sigma2 = is2.data(); p.valid = valid.data();
    p.fx = pFrame->fx; p.fy = pFrame->fy; p.cx = pFrame->cx; p.cy = pFrame->cy; p.bf = pFrame->mbf;
    p.poses7 = poses.data(); p.outlier = outl.data();
    CFSE3ObjStateOptimization(&p, bVerbose);
    for (int i = 0; i < k; i++) {                                 // mvbObjKeysOutlier is written for every edge the graph held
      const std::size_t n = vnNeedToBeOptimized[i];
      for (size_t j = 0; j < pFrame->mvpMapObjectPoints[n].size(); j++)
        if (valid[(size_t)off[i] + j]) pFrame->mvbObjKeysOutlier[n][j] = outl[(size_t)off[i] + j] != 0;
    }
    if (!p.result) return false;                                  // fewer than 15 edges (:638-639)
    for (int i = 0; i < k; i++) {                                 // :719-750
      auto* pMO = pFrame->mvMapObjects[vnNeedToBeOptimized[i]];
      const auto before = pMO->GetCFInFrameObjState(pFrame->mnId);
      auto after = before;
      after.pose = decltype(before.pose)::fromVector(&poses[(size_t)i * 7]);
      auto Swo = before;
      Swo.pose = pFrame->mSETcw.inverse() * after.pose;
      pMO->SetInFrameObjState(Swo, pFrame->mnId);
      pMO->SetCFInFrameObjState(after, pFrame->mnId);
      pMO->SetHaveBeenOptimizedInFrameFlag();
    }
    return true;
  }

  // void Optimizer::ObjectLocalBundleAdjustment(ObjectKeyFrame *pKF, const bool &bVerbose)          :755-1075
  template <class ObjectKeyFrameT>
  static void ObjectLocalBundleAdjustment(ObjectKeyFrameT* pKF, const bool& bVerbose) {
    typedef typename std::remove_pointer<typename std::decay<decltype(pKF->GetMapObjectPointMatches()[0])>::type>::type MapObjectPointT;
    const int window = 120;                                       // Optimizer.cc:47
    // local keyframes: pKF and its covisible neighbours of the last 11 object keyframes (:760-783)
    std::vector<ObjectKeyFrameT*> lLocalKeyFrames(1, pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    const int CurrentId = pKF->mnObjId;
    for (ObjectKeyFrameT* pKFi : pKF->GetVectorCovisibleKeyFrames()) {
      if (pKFi->mObjTrackId != pKF->mObjTrackId) throw std::runtime_error("ObjectLocalBundleAdjustment: neighbour of another object");   // assert(0)
      if (CurrentId - pKFi->mnObjId > 11) continue;
      pKFi->mnBALocalForKF = pKF->mnId;
      if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    // local points: everything those keyframes see (:785-801)
    std::vector<MapObjectPointT*> lLocalMapPoints;
    for (ObjectKeyFrameT* pKFi : lLocalKeyFrames)
      for (MapObjectPointT* pMP : pKFi->GetMapObjectPointMatches())
        if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
    // fixed cameras: other observers of those points inside the window (:803-819)
    std::vector<ObjectKeyFrameT*> lFixedCameras;
    for (MapObjectPointT* pMP : lLocalMapPoints)
      for (const auto& ob : pMP->GetObservations()) {
        ObjectKeyFrameT* pKFi = ob.first;
        if (CurrentId - pKFi->mnObjId > window) continue;
        if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
          pKFi->mnBAFixedForKF = pKF->mnId;
          if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
        }
      }
    // vertices (:834-858): local keyframes are VertexSE3Fix with roll / pitch locked, fixed iff mnId == 0; fixed cameras VertexSE3Expmap
    std::vector<ObjectKeyFrameT*> kfs(lLocalKeyFrames);
    kfs.insert(kfs.end(), lFixedCameras.begin(), lFixedCameras.end());
    const int np = (int)kfs.size(), nlocal = (int)lLocalKeyFrames.size(), nl = (int)lLocalMapPoints.size();
    std::vector<double> poses((size_t)np * 7), points((size_t)std::max(nl, 1) * 3, 0.0);
    std::vector<uint8_t> flags(np, 0);
    for (int i = 0; i < np; i++) {
      const pscv::Mat T = kfs[i]->GetPose();
      float m16[16];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) m16[4 * r + c] = T.template at<float>(r, c);
      ps_se3_from_mat4f(m16, &poses[(size_t)i * 7]);              // Converter::toSE3Quat(pKFi->GetPose())
      flags[i] = i < nlocal ? (uint8_t)(2 | (kfs[i]->mnId == 0 ? 1 : 0)) : (uint8_t)1;
    }
    // edges in the reference's insertion order: per local point, per observation in the order GetObservations() iterates (:876-951)
    std::vector<int32_t> e_pose, e_point;
    std::vector<float> e_obs, e_is2;
    std::vector<std::pair<ObjectKeyFrameT*, MapObjectPointT*>> e_owner;
    float fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
    for (int j = 0; j < nl; j++) {
      MapObjectPointT* pMP = lLocalMapPoints[j];
      const auto X = pMP->GetInObjFrameEigenPosition();
      for (int c = 0; c < 3; c++) points[3 * (size_t)j + c] = X(c);
      for (const auto& ob : pMP->GetObservations()) {
        ObjectKeyFrameT* pKFi = ob.first;
        if (CurrentId - pKFi->mnObjId > window || pKFi->isBad()) continue;
        const int vi = (int)(std::find(kfs.begin(), kfs.end(), pKFi) - kfs.begin());
        if (vi >= np) continue;                                    // (cannot happen: every such observer is local or fixed)
        const auto& kpUn = pKFi->mvObjKeysUn[ob.second];
        e_pose.push_back(vi); e_point.push_back(j);
        e_obs.push_back(kpUn.pt.x); e_obs.push_back(kpUn.pt.y); e_obs.push_back(pKFi->mvuObjKeysRight[ob.second]);   // uR < 0: monocular edge
        e_is2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
        e_owner.push_back(std::make_pair(pKFi, pMP));
        fx = pKFi->fx; fy = pKFi->fy; cx = pKFi->cx; cy = pKFi->cy; bf = pKFi->mbf;
      }
    }
    const int ne = (int)e_pose.size();
    std::vector<uint8_t> erase(std::max(ne, 1), 0);
    ps_ba_problem p = ps_ba_problem{};
    p.np = np; p.nl = nl; p.ne = ne; p.poses7 = poses.data(); p.pose_flags = flags.data(); p.points = points.data();
    p.e_pose = e_pose.data(); p.e_point = e_point.data(); p.e_obs = e_obs.data(); p.e_inv_sigma2 = e_is2.data();
    p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.bf = bf; p.erase = erase.data();
    if (ne > 0) ObjectLocalBundleAdjustment(&p, bVerbose);
    // write-back (:1014-1074): erase the outlier observations, keyframe poses, point positions
    for (int pass = 0; pass < 2; pass++)                          // the reference queues the monocular edges first, then the stereo ones
      for (int e = 0; e < ne; e++)
        if (erase[e] && ((e_obs[3 * (size_t)e + 2] < 0) == (pass == 0))) { e_owner[e].first->EraseMapPointMatch(e_owner[e].second); e_owner[e].second->EraseObservation(e_owner[e].first); }
    auto* pMO = pKF->mpMapObjects;
    for (int i = 0; i < nlocal; i++) {
      ObjectKeyFrameT* pKFTmp = kfs[i];
      typedef typename std::decay<decltype(pMO->GetCFInFrameObjState(0))>::type ObjectStateT;
      ObjectStateT x;
      x.pose = decltype(x.pose)::fromVector(&poses[(size_t)i * 7]);
      x.scale = pKFTmp->mScale;
      pKFTmp->SetPose(x.pose);
      pMO->SetCFObjectKeyFrameObjState(pKFTmp, x);
      pMO->SetCFInFrameObjState(x, pKFTmp->mnFrameId);
    }
    for (int j = 0; j < nl; j++) {
      pscv::Mat X(3, 1, pscv::CV_32F);
      for (int c = 0; c < 3; c++) X.template at<float>(c) = (float)points[3 * (size_t)j + c];   // Converter::toCvMat(Vector3d)
      lLocalMapPoints[j]->SetInObjFramePosition(X);
      lLocalMapPoints[j]->UpdateNormalAndDepth();
    }
  }
};

}  // namespace ORB_SLAM2
