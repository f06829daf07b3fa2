// ORB_SLAM2::ORBextractor with the reference's signature (/root/reference/include/ORBextractor.h:51-85), implemented
// on the C-ABI of libpointslot_hip.so.  A Frame.cc / Tracking.cc-shaped caller compiles unchanged:
//   ORBextractor ex(2000, 1.2f, 8, 20, 5);  ex(image, cv::Mat(), keypoints, descriptors);  ex.mvImagePyramid[l]
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "slotcv.h"

namespace ORB_SLAM2 {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
      : nfeatures_(nfeatures), nlevels_(nlevels), scaleFactor_(scaleFactor) {
    ps_orb_config cfg{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, 1, device};
    if (ps_orb_create(&cfg, &h_) != PS_OK) throw std::runtime_error(std::string("ps_orb_create: ") + ps_last_error());
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    ps_orb_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), nullptr);
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { ps_orb_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image; the mask is ignored (as in the reference).
  void operator()(const pscv::Mat& image, const pscv::Mat& /*mask*/, std::vector<pscv::KeyPoint>& keypoints,
                  pscv::Mat& descriptors) {
    if (image.empty()) return;                                   // ORBextractor.cc:1046-1047
    if (image.type() != 0 /*CV_8UC1*/) throw std::runtime_error("image.type() == CV_8UC1");   // :1050 (assert)
    const int cap = nfeatures_ + 4 * nlevels_ + 64;
    keypoints.resize(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    // mvImagePyramid: padded parent buffers owned here, ROI views handed out (ORBextractor.cc:1113-1115).  A caller that
    // matches stereo pairs on the device (ps_orb_stereo_match_pair) clears mbDownloadPyramid and saves the 1.7 MB read-back.
    std::vector<uint8_t*> planes(nlevels_, nullptr);
    parents_.resize(nlevels_);
    for (int l = 0; mbDownloadPyramid && l < nlevels_; l++) {
      int32_t wl, hl;
      ps_orb_level_size(h_, image.cols, image.rows, l, &wl, &hl);
      parents_[l].create(hl + 38, wl + 38, 0);
      planes[l] = parents_[l].data;
      mvImagePyramid[l] = pscv::Mat(hl, wl, 0, parents_[l].data + 19 * parents_[l].step + 19, parents_[l].step);
    }
    int n = 0;
    static_assert(sizeof(pscv::KeyPoint) == sizeof(ps_keypoint), "KeyPoint layout");
    if (ps_orb_extract(h_, image.data, image.cols, image.rows, (int)image.step, (ps_keypoint*)keypoints.data(), desc.data(),
                       cap, &n, mbDownloadPyramid ? planes.data() : nullptr) != PS_OK)
      throw std::runtime_error(std::string("ps_orb_extract: ") + ps_last_error());
    keypoints.resize(n);
    if (n == 0) { descriptors.release(); return; }               // :1064-1065
    descriptors.create(n, 32, 0);
    std::memcpy(descriptors.data, desc.data(), (size_t)n * 32);
  }

  int inline GetLevels() { return nlevels_; }
  float inline GetScaleFactor() { return scaleFactor_; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  std::vector<pscv::Mat> mvImagePyramid;   // public in the reference: Frame::ComputeStereoMatches reads it
  bool mbDownloadPyramid = true;           // not in the reference: see operator()
  ps_orb* handle() { return h_; }          // for the device-resident stereo matcher

 protected:
  ps_orb* h_ = nullptr;
  int nfeatures_, nlevels_;
  float scaleFactor_;
  std::vector<pscv::Mat> parents_;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2
