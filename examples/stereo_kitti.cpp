// Usage: stereo_kitti path_to_sequence [max_frames]        (cf. /root/reference/Examples/Stereo/stereo_kitti.cc:55-166)
//
// The reference's stereo driver shape on the C++ host shim: loads a sequence in the reference's on-disk layout
// (image_02 / image_03 / timestamp.txt; images as binary PGM because this build image has no OpenCV to decode PNG — a
// maintainer's build defines POINTSLOT_WITH_OPENCV and uses cv::imread), tracks it with ORB_SLAM2::StereoOdometry, and
// writes CameraTrajectory.txt in the System::SaveTrajectoryKITTI format (src/System.cc:395-402).
//   g++ -std=c++17 -O2 -pthread -I pointslot_amd/host -I include examples/stereo_kitti.cpp -L pointslot_amd -lpointslot_hip
#include <chrono>
#include <cstdio>
#include "StereoOdometry.h"
#include "kitti_io.h"

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "Usage: ./stereo_kitti path_to_sequence [max_frames]" << std::endl; return 1; }
  const std::string seq = argv[1];
  std::vector<std::string> vstrImageLeft, vstrImageRight;
  std::vector<double> vTimestamps;
  LoadImages(seq, vstrImageLeft, vstrImageRight, vTimestamps);
  int nImages = (int)vstrImageLeft.size();
  if (argc > 2) nImages = std::min(nImages, std::atoi(argv[2]));
  if (nImages == 0) { std::cerr << "no timestamp.txt / images under " << seq << std::endl; return 1; }
  std::map<std::string, double> calib = LoadCalib(seq);
  std::vector<unsigned char> imLeft, imRight;
  int w = 0, h = 0, wr = 0, hr = 0;
  if (!LoadPGM(vstrImageLeft[0], imLeft, w, h)) { std::cerr << "Failed to load image at: " << vstrImageLeft[0] << std::endl; return 1; }
  ORB_SLAM2::StereoOdometry SLAM((float)calib["Camera.fx"], (float)calib["Camera.fy"], (float)calib["Camera.cx"], (float)calib["Camera.cy"],
                                 (float)calib["Camera.bf"], w, h, (float)calib["ThDepth"]);
  std::vector<float> vTimesTrack(nImages);
  // where the time goes: the extractors' kernels by HIP events on their own streams, the matcher's / optimiser's kernels per call
  ps_orb_enable_stage_timing(SLAM.leftExtractor().handle(), 1);
  ps_orb_enable_stage_timing(SLAM.rightExtractor().handle(), 1);
  std::cout << "Start processing sequence ... Images in the sequence: " << nImages << std::endl;
  for (int ni = 0; ni < nImages; ni++) {
    if (!LoadPGM(vstrImageLeft[ni], imLeft, w, h) || !LoadPGM(vstrImageRight[ni], imRight, wr, hr) || wr != w || hr != h) {
      std::cerr << "Failed to load image at: " << vstrImageLeft[ni] << std::endl;
      return 1;
    }
    pscv::Mat L(h, w, 0, imLeft.data()), R(h, w, 0, imRight.data());
    const auto t1 = std::chrono::steady_clock::now();
    if (ni == 1) SLAM.split.clear();      // the first frame initialises (and pays the first launches): not part of the split, like the median below
    const bool ok = SLAM.Track(L, R);
    const auto t2 = std::chrono::steady_clock::now();
    vTimesTrack[ni] = (float)std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
    std::printf("frame %d: %s matches %d map %d local inliers %d\n", ni, ok ? "ok" : (SLAM.state == ORB_SLAM2::StereoOdometry::NOT_INITIALIZED ? "not initialised" : "LOST"),
                SLAM.lastMatches, SLAM.lastMapMatches, SLAM.lastLocalInliers);
  }
  std::vector<float> sorted(vTimesTrack.begin() + (nImages > 1 ? 1 : 0), vTimesTrack.end());
  std::sort(sorted.begin(), sorted.end());
  float total = 0;
  for (float t : sorted) total += t;
  std::cout << "-------" << std::endl;
  std::cout << "median tracking time: " << 1e3 * sorted[sorted.size() / 2] << " ms" << std::endl;
  std::cout << "mean tracking time: " << 1e3 * total / sorted.size() << " ms" << std::endl;
  {
    // the split SURVEY section 7 asks for, per frame (frames 1 ..): wall time inside the C-ABI calls by kind, the kernels' own time, the rest
    const ORB_SLAM2::StereoOdometry::Split& sp = SLAM.split;
    const double f = sp.frames > 0 ? 1e3 / (double)sp.frames : 0.0;
    double orbKernelMs = 0;
    for (ORB_SLAM2::ORBextractor* ex : {&SLAM.leftExtractor(), &SLAM.rightExtractor()}) {
      const char* names[16]; float ms[16]; int n = 0;
      double one = 0;
      if (ps_orb_stage_times(ex->handle(), names, ms, 16, &n) == PS_OK) for (int i = 0; i < n; i++) one += ms[i];
      orbKernelMs = std::max(orbKernelMs, one);     // the two extractors run side by side on two streams: the longer one is on the frame's path
    }
    const double wall = (sp.extract + sp.stereo + sp.search + sp.pose + sp.host) * f;
    std::printf("split_json: {\"frames\": %ld, \"wall_ms\": %.4f, \"extract_call_ms\": %.4f, \"extract_kernels_ms\": %.4f, \"stereo_call_ms\": %.4f, "
                "\"search_call_ms\": %.4f, \"search_kernels_ms\": %.4f, \"search_calls_per_frame\": %.3f, \"pose_call_ms\": %.4f, \"pose_kernels_ms\": %.4f, "
                "\"pose_calls_per_frame\": %.3f, \"host_marshalling_ms\": %.4f}\n",
                sp.frames, wall, sp.extract * f, orbKernelMs, sp.stereo * f, sp.search * f, sp.searchKernelMs / std::max(1L, sp.frames),
                (double)sp.searchCalls / std::max(1L, sp.frames), sp.pose * f, sp.poseKernelMs / std::max(1L, sp.frames),
                (double)sp.poseCalls / std::max(1L, sp.frames), sp.host * f);
  }
  SaveTrajectoryKITTI(seq + "/CameraTrajectory.txt", SLAM.trajectory);
  std::cout << std::endl << "trajectory saved!" << std::endl;
  return 0;
}
