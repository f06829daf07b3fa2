// The object-feature detector of the reference, Frame.cc:2623-2627:
//   void OpencvORBDetector(cv::Mat im, cv::Mat ObjMask, vector<cv::KeyPoint> &kp, cv::Mat &descriptor) {
//     cv::Ptr<cv::FeatureDetector> detector = cv::ORB::create(1000, 1.2, 8, 19);
//     detector->detectAndCompute(im, ObjMask, kp, descriptor); }
// with the same name and signature on the C-ABI: ps_cvorb_* is OpenCV 3.4.3's ORB_Impl::detectAndCompute restated for the GPU
// (unverifiable against OpenCV in the build image, see include/pointslot_hip.h).
// PS_OBJECT_ORB_STAND_IN selects the earlier stand-in instead (ps_orb_extract_masked: THIS library's extractor with the keypoints
// outside the mask dropped before the quadtree) - a different keypoint set, kept for comparison.
#pragma once
#include <memory>
#include "ORBextractor.h"

namespace ORB_SLAM2 {

// one detector handle per calling thread (the reference runs the left and the right image on two threads, Frame.cc:2650-2653)
inline void OpencvORBDetector(pscv::Mat im, pscv::Mat ObjMask, std::vector<pscv::KeyPoint>& kp, pscv::Mat& descriptor, int device = 0) {
  kp.clear();
  if (im.empty()) { descriptor.release(); return; }
  if (im.type() != 0 || (!ObjMask.empty() && (ObjMask.type() != 0 || ObjMask.rows != im.rows || ObjMask.cols != im.cols)))
    throw std::runtime_error("OpencvORBDetector: image and mask must be CV_8UC1 of the same size");
  const int cap = 16 * 1000 + 4096;   // retainBest keeps ties: the count may exceed nfeatures
  kp.resize(cap);
  std::vector<uint8_t> desc((size_t)cap * 32);
  int n = 0;
#ifdef PS_OBJECT_ORB_STAND_IN
  thread_local std::unique_ptr<ORBextractor> ex;
  if (!ex) ex.reset(new ORBextractor(1000, 1.2f, 8, 20, 5, device));
  const int rc = ps_orb_extract_masked(ex->handle(), im.data, ObjMask.empty() ? nullptr : ObjMask.data, im.cols, im.rows, (int)im.step, (int)ObjMask.step,
                                       (ps_keypoint*)kp.data(), desc.data(), cap, &n);
#else
  struct Handle { ps_cvorb* h = nullptr; ~Handle() { ps_cvorb_destroy(h); } };
  thread_local Handle det;
  if (!det.h && ps_cvorb_create(1000, 1.2f, 8, 19, 20, device, &det.h) != PS_OK) throw std::runtime_error(std::string("ps_cvorb_create: ") + ps_last_error());
  const int rc = ps_cvorb_detect_and_compute(det.h, im.data, ObjMask.empty() ? nullptr : ObjMask.data, im.cols, im.rows, (int)im.step, (int)ObjMask.step,
                                             (ps_keypoint*)kp.data(), desc.data(), cap, &n);
#endif
  if (rc != PS_OK) throw std::runtime_error(std::string("OpencvORBDetector: ") + ps_last_error());
  kp.resize(n);
  if (n == 0) { descriptor.release(); return; }
  descriptor.create(n, 32, 0);
  std::memcpy(descriptor.data, desc.data(), (size_t)n * 32);
}

}  // namespace ORB_SLAM2
