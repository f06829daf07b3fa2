// C++ smoke of the host shim classes (pointslot_amd/host/): an ORB_SLAM2-shaped caller, compiled with g++ against
// libpointslot_hip.so, no OpenCV.  Usage: shim_smoke <raw u8 image> <w> <h> <out prefix>
// Writes <prefix>.kps (28-byte records) and <prefix>.desc for the Python test to compare with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"
#include "ObjectORB.h"

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage\n"); return 2; }
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<unsigned char> px((size_t)w * h);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(px.data(), 1, px.size(), f) != px.size()) { std::fprintf(stderr, "cannot read image\n"); return 2; }
  std::fclose(f);
  pscv::Mat image(h, w, 0, px.data());
  ORB_SLAM2::ORBextractor* mpORBextractorLeft = new ORB_SLAM2::ORBextractor(1000, 1.2f, 8, 20, 5);   // Tracking.cc:385
  std::vector<pscv::KeyPoint> mvKeys;
  pscv::Mat mDescriptors;
  (*mpORBextractorLeft)(image, pscv::Mat(), mvKeys, mDescriptors);                                      // Frame.cc:1658
  std::printf("keypoints %zu levels %d scale[7] %.6f pyramid0 %dx%d\n", mvKeys.size(), mpORBextractorLeft->GetLevels(),
              mpORBextractorLeft->GetScaleFactors()[7], mpORBextractorLeft->mvImagePyramid[0].cols,
              mpORBextractorLeft->mvImagePyramid[0].rows);
  if (mvKeys.empty() || mpORBextractorLeft->mvImagePyramid[0].at<unsigned char>(5, 7) != px[5 * w + 7]) return 1;
  std::string pre = argv[4];
  f = std::fopen((pre + ".kps").c_str(), "wb"); std::fwrite(mvKeys.data(), 28, mvKeys.size(), f); std::fclose(f);
  f = std::fopen((pre + ".desc").c_str(), "wb"); std::fwrite(mDescriptors.data, 32, mvKeys.size(), f); std::fclose(f);
  // DescriptorDistance on the host + the bulk matrix on the GPU agree
  ORB_SLAM2::ORBmatcher matcher(0.9, true);
  std::vector<uint16_t> D;
  matcher.DescriptorDistanceMatrix(mDescriptors, mDescriptors, D);
  const int n = (int)mvKeys.size();
  for (int i = 0; i < n; i += 97)
    for (int j = 0; j < n; j += 89)
      if (D[(size_t)i * n + j] != ORB_SLAM2::ORBmatcher::DescriptorDistance(mDescriptors.row(i), mDescriptors.row(j))) return 1;
  (void)ORB_SLAM2::Optimizer::handle(0);
  // Frame::ExtractObjORB -> OpencvORBDetector(left, LeftObjMask, keypoints_1, descriptors_1) (Frame.cc:2623-2651): the left half masked
  pscv::Mat objMask(h, w, 0);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) objMask.at<unsigned char>(y, x) = (x > w / 8 && x < w / 2 && y > h / 6) ? 255 : 0;
  std::vector<pscv::KeyPoint> keypoints_1;
  pscv::Mat descriptors_1;
  ORB_SLAM2::OpencvORBDetector(image, objMask, keypoints_1, descriptors_1);
  std::printf("object keypoints %zu\n", keypoints_1.size());
  if (keypoints_1.empty()) return 1;
  f = std::fopen((pre + ".okps").c_str(), "wb"); std::fwrite(keypoints_1.data(), 28, keypoints_1.size(), f); std::fclose(f);
  f = std::fopen((pre + ".odesc").c_str(), "wb"); std::fwrite(descriptors_1.data, 32, keypoints_1.size(), f); std::fclose(f);
  delete mpORBextractorLeft;
  std::printf("shim smoke ok\n");
  return 0;
}
