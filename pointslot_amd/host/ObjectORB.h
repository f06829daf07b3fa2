// The object-feature detector of the reference, Frame.cc:2623-2627:
//   void OpencvORBDetector(cv::Mat im, cv::Mat ObjMask, vector<cv::KeyPoint> &kp, cv::Mat &descriptor) {
//     cv::Ptr<cv::FeatureDetector> detector = cv::ORB::create(1000, 1.2, 8, 19);
//     detector->detectAndCompute(im, ObjMask, kp, descriptor); }
// with the same name and signature on the C-ABI.  SURVEY.md 8f-2: cv::ORB is a different extractor (Harris-ranked, its own
// pyramid) that lives in un-vendored OpenCV; what runs here is the DECLARED STAND-IN ps_orb_extract_masked - this library's
// extractor (1000 features, 1.2, 8 levels, FAST 20 / 5, edge threshold 19) with the keypoints outside the mask dropped before
// the quadtree.  Object keypoint sets therefore differ from a true PointSLOT run even where the static side is identical.
#pragma once
#include <memory>
#include <mutex>
#include "ORBextractor.h"

namespace ORB_SLAM2 {

// one extractor handle per calling thread (the reference runs the left and the right image on two threads, Frame.cc:2650-2653)
inline void OpencvORBDetector(pscv::Mat im, pscv::Mat ObjMask, std::vector<pscv::KeyPoint>& kp, pscv::Mat& descriptor, int device = 0) {
  thread_local std::unique_ptr<ORBextractor> ex;
  if (!ex) ex.reset(new ORBextractor(1000, 1.2f, 8, 20, 5, device));
  kp.clear();
  if (im.empty()) { descriptor.release(); return; }
  if (im.type() != 0 || (!ObjMask.empty() && (ObjMask.type() != 0 || ObjMask.rows != im.rows || ObjMask.cols != im.cols)))
    throw std::runtime_error("OpencvORBDetector: image and mask must be CV_8UC1 of the same size");
  const int cap = 1000 + 4 * 8 + 64;
  kp.resize(cap);
  std::vector<uint8_t> desc((size_t)cap * 32);
  int n = 0;
  if (ps_orb_extract_masked(ex->handle(), im.data, ObjMask.empty() ? nullptr : ObjMask.data, im.cols, im.rows, (int)im.step, (int)ObjMask.step,
                            (ps_keypoint*)kp.data(), desc.data(), cap, &n) != PS_OK)
    throw std::runtime_error(std::string("ps_orb_extract_masked: ") + ps_last_error());
  kp.resize(n);
  if (n == 0) { descriptor.release(); return; }
  descriptor.create(n, 32, 0);
  std::memcpy(descriptor.data, desc.data(), (size_t)n * 32);
}

}  // namespace ORB_SLAM2
