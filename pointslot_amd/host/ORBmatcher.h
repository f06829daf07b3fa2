// ORB_SLAM2::ORBmatcher hot members (/root/reference/include/ORBmatcher.h:47-118) on the C-ABI.  The reference's
// methods take Frame& and write MapPoint* into it; Frame / MapPoint are outside the hot path and are not rebuilt, so the
// shim takes views of exactly the Frame fields each method reads (INTEGRATION.md lists the field mapping) and
// returns index assignments that the caller turns back into pointer writes.
#pragma once
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
  static float RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }   // ORBmatcher.cc:252-258

 protected:
  int run(ps_proj_problem& p) {
    if (ps_search_by_projection(h_, &p, 1) != PS_OK) throw std::runtime_error(ps_last_error());
    return p.nmatches;
  }
  ps_matcher* h_ = nullptr;
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2
