// Sequence I/O shared by the example drivers: the reference's on-disk layout (image_02 / image_03 / timestamp.txt,
// /root/reference/Examples/Stereo/stereo_kitti.cc:170-230; images as binary PGM because this build image has no OpenCV to
// decode PNG), "key: value" calibration lines, and the System::SaveTrajectoryKITTI line format (src/System.cc:395-402).
#pragma once
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

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


// calibration: "key: value" lines (the reference reads them from the settings yaml)
static std::map<std::string, double> LoadCalib(const std::string& seq) {
  std::map<std::string, double> calib = {{"Camera.fx", 721.5377}, {"Camera.fy", 721.5377}, {"Camera.cx", 609.5593}, {"Camera.cy", 172.854},
                                         {"Camera.bf", 384.38148}, {"ThDepth", 35}};
  std::ifstream fc((seq + "/calib.txt").c_str());
  std::string line;
  while (std::getline(fc, line)) {
    const size_t k = line.find(':');
    if (k != std::string::npos) calib[line.substr(0, k)] = std::atof(line.c_str() + k + 1);
  }
  return calib;
}

// System::SaveTrajectoryKITTI line format: row-major 3x4 [Rwc | twc] per tracked frame
static void SaveTrajectoryKITTI(const std::string& path, const std::vector<std::vector<float>>& trajectory) {
  std::ofstream f(path.c_str());
  f << std::fixed;
  for (const std::vector<float>& T : trajectory) {
    if (T.empty()) continue;
    float Rwc[9], twc[3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rwc[3 * r + c] = T[4 * c + r];
    for (int r = 0; r < 3; r++) twc[r] = -(Rwc[3 * r] * T[3] + Rwc[3 * r + 1] * T[7] + Rwc[3 * r + 2] * T[11]);
    f << std::setprecision(9) << Rwc[0] << " " << Rwc[1] << " " << Rwc[2] << " " << twc[0] << " " << Rwc[3] << " " << Rwc[4] << " " << Rwc[5] << " "
      << twc[1] << " " << Rwc[6] << " " << Rwc[7] << " " << Rwc[8] << " " << twc[2] << std::endl;
  }
}
