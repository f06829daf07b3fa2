// ORB_SLAM2::Optimizer hot static members (/root/reference/include/Optimizer.h:51-61) on the C-ABI.  The reference's
// functions take Frame* / ObjectKeyFrame* and mutate them; the shim takes the arrays those functions read
// (ps_pose_problem / ps_cfse3_problem / ps_ba_problem, field mapping in include/pointslot_hip.h and INTEGRATION.md) and
// returns the same values the reference returns.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/pointslot_hip.h"

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
};

}  // namespace ORB_SLAM2
