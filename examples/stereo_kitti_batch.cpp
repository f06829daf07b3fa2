// Usage: stereo_kitti_batch [--max-frames N] [--device D] path_to_sequence_1 path_to_sequence_2 ...
//
// BASELINE config 4 on one GPU: several independent stereo sequences (the reference's on-disk layout, see kitti_io.h) tracked
// in lockstep by ORB_SLAM2::StereoOdometryBatch — every step extracts all sequences' stereo pairs as one batch and serves
// each round of SearchByProjection / PoseOptimization problems with one C-ABI call.  All images are loaded into page-locked
// memory before the clock starts (the figure is tracking throughput, not disk or PGM decoding).  Writes
// CameraTrajectoryBatch.txt next to each sequence (System::SaveTrajectoryKITTI format) and prints one JSON line.
//   g++ -std=c++17 -O2 -pthread -I pointslot_amd/host -I include examples/stereo_kitti_batch.cpp -L pointslot_amd -lpointslot_hip
#include <chrono>
#include <cstdio>
#include "StereoOdometry.h"
#include "kitti_io.h"

int main(int argc, char** argv) {
  std::vector<std::string> seqs;
  int maxFrames = 1 << 30, device = 0;
  for (int a = 1; a < argc; a++) {
    if (std::string(argv[a]) == "--max-frames" && a + 1 < argc) maxFrames = std::atoi(argv[++a]);
    else if (std::string(argv[a]) == "--device" && a + 1 < argc) device = std::atoi(argv[++a]);
    else seqs.push_back(argv[a]);
  }
  if (seqs.empty()) { std::cerr << "Usage: ./stereo_kitti_batch [--max-frames N] [--device D] path_to_sequence ..." << std::endl; return 1; }
  const int S = (int)seqs.size();
  std::vector<std::vector<std::string>> vstrLeft(S), vstrRight(S);
  int nImages = maxFrames;
  for (int k = 0; k < S; k++) {
    std::vector<double> vTimestamps;
    LoadImages(seqs[k], vstrLeft[k], vstrRight[k], vTimestamps);
    nImages = std::min(nImages, (int)vstrLeft[k].size());
  }
  if (nImages <= 0) { std::cerr << "no timestamp.txt / images under the given sequences" << std::endl; return 1; }
  std::map<std::string, double> calib = LoadCalib(seqs[0]);
  // all frames of all sequences into page-locked memory
  std::vector<unsigned char> px;
  int w = 0, h = 0;
  if (!LoadPGM(vstrLeft[0][0], px, w, h)) { std::cerr << "Failed to load image at: " << vstrLeft[0][0] << std::endl; return 1; }
  const size_t pitch = (size_t)w * h;
  unsigned char* pinned = (unsigned char*)ps_pinned_alloc(pitch * 2 * (size_t)S * nImages);
  if (!pinned) { std::cerr << ps_last_error() << std::endl; return 1; }
  auto image = [&](int k, int ni, int right) { return pinned + pitch * ((((size_t)ni * S) + k) * 2 + right); };
  for (int k = 0; k < S; k++)
    for (int ni = 0; ni < nImages; ni++)
      for (int right = 0; right < 2; right++) {
        int wi = 0, hi = 0;
        const std::string& path = right ? vstrRight[k][ni] : vstrLeft[k][ni];
        if (!LoadPGM(path, px, wi, hi) || wi != w || hi != h) { std::cerr << "Failed to load image at: " << path << std::endl; return 1; }
        std::memcpy(image(k, ni, right), px.data(), pitch);
      }
  try {
    ORB_SLAM2::StereoOdometryBatch SLAM(S, (float)calib["Camera.fx"], (float)calib["Camera.fy"], (float)calib["Camera.cx"], (float)calib["Camera.cy"],
                                        (float)calib["Camera.bf"], w, h, (float)calib["ThDepth"], 2000, 1.2f, 8, 20, 5, device);
    const bool prefetch = std::getenv("PS_ODO_NO_PREFETCH") == nullptr;   // queue step n+1's extraction during step n's search / pose rounds
    std::vector<double> vTimesTrack(nImages);
    std::vector<std::vector<double>> vParts(4, std::vector<double>(nImages));   // frames / host / search / pose seconds per step
    std::vector<const uint8_t*> left(S), right(S), nextLeft(S), nextRight(S);
    int lost = 0;
    std::cout << "Start processing " << S << " sequences in lockstep ... Images per sequence: " << nImages << std::endl;
    for (int ni = 0; ni < nImages; ni++) {
      const bool more = prefetch && ni + 1 < nImages;
      for (int k = 0; k < S; k++) {
        left[k] = image(k, ni, 0); right[k] = image(k, ni, 1);
        if (more) { nextLeft[k] = image(k, ni + 1, 0); nextRight[k] = image(k, ni + 1, 1); }
      }
      const double e0 = SLAM.tExtract, h0 = SLAM.tHost, s0 = SLAM.tSearch, p0 = SLAM.tPose;
      const auto t1 = std::chrono::steady_clock::now();
      const int tracked = SLAM.TrackAll(left, right, w, more ? &nextLeft : nullptr, more ? &nextRight : nullptr);
      const auto t2 = std::chrono::steady_clock::now();
      vTimesTrack[ni] = std::chrono::duration<double>(t2 - t1).count();
      lost += S - tracked;
      vParts[0][ni] = SLAM.tExtract - e0; vParts[1][ni] = SLAM.tHost - h0; vParts[2][ni] = SLAM.tSearch - s0; vParts[3][ni] = SLAM.tPose - p0;
      std::printf("step %d: %d of %d sequences tracked, %.3f ms (frames %.3f, host %.3f, search %.3f, pose %.3f)\n", ni, tracked, S, 1e3 * vTimesTrack[ni],
                  1e3 * vParts[0][ni], 1e3 * vParts[1][ni], 1e3 * vParts[2][ni], 1e3 * vParts[3][ni]);
    }
    for (int k = 0; k < S; k++) SaveTrajectoryKITTI(seqs[k] + "/CameraTrajectoryBatch.txt", SLAM.sequence(k).trajectory);
    // the first step initialises every sequence (and builds the plans): statistics over the tracked steps
    std::vector<double> sorted(vTimesTrack.begin() + (nImages > 1 ? 1 : 0), vTimesTrack.end());
    std::sort(sorted.begin(), sorted.end());
    double total = 0;
    for (double t : sorted) total += t;
    const double median = sorted[sorted.size() / 2], mean = total / sorted.size();
    double part[4];
    for (int q = 0; q < 4; q++) {
      std::vector<double> v(vParts[q].begin() + (nImages > 1 ? 1 : 0), vParts[q].end());
      std::sort(v.begin(), v.end());
      part[q] = v[v.size() / 2];
    }
    std::printf("{\"sequences\": %d, \"frames_per_sequence\": %d, \"untracked_frames\": %d, \"median_ms_per_step\": %.4f, \"mean_ms_per_step\": %.4f, "
                "\"frames_per_s\": %.1f, \"ms_per_step_frames\": %.4f, \"ms_per_step_host\": %.4f, \"ms_per_step_search\": %.4f, "
                "\"ms_per_step_pose\": %.4f, \"device_rounds_per_step\": %.2f}\n",
                S, nImages, lost, 1e3 * median, 1e3 * mean, S / median, 1e3 * part[0], 1e3 * part[1], 1e3 * part[2], 1e3 * part[3],
                SLAM.rounds / (double)nImages);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    ps_pinned_free(pinned);
    return 2;
  }
  ps_pinned_free(pinned);
  return 0;
}
