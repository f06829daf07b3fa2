// Usage: stereo_kitti path_to_sequence [max_frames]        (cf. /root/reference/Examples/Stereo/stereo_kitti.cc:55-166)
//
// The reference's stereo driver shape on the C++ host shim: loads a sequence in the reference's on-disk layout
// (image_02 / image_03 / timestamp.txt; images as binary PGM because this build image has no OpenCV to decode PNG — a
// maintainer's build defines POINTSLOT_WITH_OPENCV and uses cv::imread), tracks it with ORB_SLAM2::StereoOdometry, and
// writes CameraTrajectory.txt in the System::SaveTrajectoryKITTI format (src/System.cc:395-402).
//   g++ -std=c++17 -O2 -pthread -I pointslot_amd/host -I include examples/stereo_kitti.cpp -L pointslot_amd -lpointslot_hip
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "StereoOdometry.h"

static bool LoadPGM(const std::string& path, std::vector<unsigned char>& px, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  std::string magic;
  int maxv = 0;
  if (!(f >> magic) || magic != "P5") return false;
  auto skip = [&]() { while (f >> std::ws && f.peek() == '#') { std::string line; std::getline(f, line); } };
  skip(); f >> w; skip(); f >> h; skip(); f >> maxv;
  f.get();
  if (!f || maxv != 255 || w <= 0 || h <= 0) return false;
  px.resize((size_t)w * h);
  f.read((char*)px.data(), (std::streamsize)px.size());
  return (bool)f;
}

static void LoadImages(const std::string& strPathToSequence, std::vector<std::string>& vstrImageLeft,
                       std::vector<std::string>& vstrImageRight, std::vector<double>& vTimestamps) {   // stereo_kitti.cc:170-230
  std::ifstream fTimes((strPathToSequence + "/timestamp.txt").c_str());
  std::string s;
  while (std::getline(fTimes, s)) {
    if (s.empty()) continue;
    std::stringstream ss(s);
    double t;
    ss >> t;
    vTimestamps.push_back(t);
  }
  const int nTimes = (int)vTimestamps.size();
  vstrImageLeft.resize(nTimes); vstrImageRight.resize(nTimes);
  for (int i = 0; i < nTimes; i++) {
    std::stringstream ss;
    ss << std::setfill('0') << std::setw(6) << i;
    vstrImageLeft[i] = strPathToSequence + "/image_02/" + ss.str() + ".pgm";
    vstrImageRight[i] = strPathToSequence + "/image_03/" + ss.str() + ".pgm";
  }
}

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "Usage: ./stereo_kitti path_to_sequence [max_frames]" << std::endl; return 1; }
  const std::string seq = argv[1];
  std::vector<std::string> vstrImageLeft, vstrImageRight;
  std::vector<double> vTimestamps;
  LoadImages(seq, vstrImageLeft, vstrImageRight, vTimestamps);
  int nImages = (int)vstrImageLeft.size();
  if (argc > 2) nImages = std::min(nImages, std::atoi(argv[2]));
  if (nImages == 0) { std::cerr << "no timestamp.txt / images under " << seq << std::endl; return 1; }
  // calibration: "key: value" lines (the reference reads them from the settings yaml)
  std::map<std::string, double> calib = {{"Camera.fx", 721.5377}, {"Camera.fy", 721.5377}, {"Camera.cx", 609.5593}, {"Camera.cy", 172.854},
                                         {"Camera.bf", 384.38148}, {"ThDepth", 35}};
  {
    std::ifstream fc((seq + "/calib.txt").c_str());
    std::string line;
    while (std::getline(fc, line)) {
      const size_t k = line.find(':');
      if (k != std::string::npos) calib[line.substr(0, k)] = std::atof(line.c_str() + k + 1);
    }
  }
  std::vector<unsigned char> imLeft, imRight;
  int w = 0, h = 0, wr = 0, hr = 0;
  if (!LoadPGM(vstrImageLeft[0], imLeft, w, h)) { std::cerr << "Failed to load image at: " << vstrImageLeft[0] << std::endl; return 1; }
  ORB_SLAM2::StereoOdometry SLAM((float)calib["Camera.fx"], (float)calib["Camera.fy"], (float)calib["Camera.cx"], (float)calib["Camera.cy"],
                                 (float)calib["Camera.bf"], w, h, (float)calib["ThDepth"]);
  std::vector<float> vTimesTrack(nImages);
  std::cout << "Start processing sequence ... Images in the sequence: " << nImages << std::endl;
  for (int ni = 0; ni < nImages; ni++) {
    if (!LoadPGM(vstrImageLeft[ni], imLeft, w, h) || !LoadPGM(vstrImageRight[ni], imRight, wr, hr) || wr != w || hr != h) {
      std::cerr << "Failed to load image at: " << vstrImageLeft[ni] << std::endl;
      return 1;
    }
    pscv::Mat L(h, w, 0, imLeft.data()), R(h, w, 0, imRight.data());
    const auto t1 = std::chrono::steady_clock::now();
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
  // System::SaveTrajectoryKITTI line format
  std::ofstream f((seq + "/CameraTrajectory.txt").c_str());
  f << std::fixed;
  for (const std::vector<float>& T : SLAM.trajectory) {
    if (T.empty()) continue;
    float Rwc[9], twc[3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rwc[3 * r + c] = T[4 * c + r];
    for (int r = 0; r < 3; r++) twc[r] = -(Rwc[3 * r] * T[3] + Rwc[3 * r + 1] * T[7] + Rwc[3 * r + 2] * T[11]);
    f << std::setprecision(9) << Rwc[0] << " " << Rwc[1] << " " << Rwc[2] << " " << twc[0] << " " << Rwc[3] << " " << Rwc[4] << " " << Rwc[5] << " "
      << twc[1] << " " << Rwc[6] << " " << Rwc[7] << " " << Rwc[8] << " " << twc[2] << std::endl;
  }
  f.close();
  std::cout << std::endl << "trajectory saved!" << std::endl;
  return 0;
}
