// Parity driver for the device-resident lockstep tracker (ps_tracker_*): the same generated sequences through
//   (a) ORB_SLAM2::StereoOdometryBatch — the host-driven chain over the per-call C-ABI (itself held to the CPU checker by
//       tests/test_tracker_gpu.py / test_stereo_kitti_cpp.py), and
//   (b) ORB_SLAM2::StereoOdometryDevice — one ps_tracker_step per frame, everything on the device,
// and compares every frame of every sequence: tracked flag, Tcw bit for bit, the match / inlier counts of Tracking::Track.
// Usage: track_device_check [--max-frames N] seq_dir ...   (prints one JSON line; exit code 0 also when they differ)
#include <cstdio>
#include <cstring>
#include "StereoOdometry.h"
#include "../../examples/kitti_io.h"

int main(int argc, char** argv) {
  std::vector<std::string> seqs;
  int maxFrames = 1 << 30;
  for (int a = 1; a < argc; a++) {
    if (std::string(argv[a]) == "--max-frames" && a + 1 < argc) maxFrames = std::atoi(argv[++a]);
    else seqs.push_back(argv[a]);
  }
  if (seqs.empty()) { std::cerr << "Usage: track_device_check [--max-frames N] seq_dir ..." << std::endl; return 1; }
  const int S = (int)seqs.size();
  std::vector<std::vector<std::string>> L(S), R(S);
  int nImages = maxFrames;
  for (int k = 0; k < S; k++) { std::vector<double> ts; LoadImages(seqs[k], L[k], R[k], ts); nImages = std::min(nImages, (int)L[k].size()); }
  std::map<std::string, double> calib = LoadCalib(seqs[0]);
  std::vector<unsigned char> px;
  int w = 0, h = 0;
  if (nImages <= 0 || !LoadPGM(L[0][0], px, w, h)) { std::cerr << "cannot load " << seqs[0] << std::endl; return 1; }
  const size_t pitch = (size_t)w * h;
  unsigned char* pinned = (unsigned char*)ps_pinned_alloc(pitch * 2 * (size_t)S * nImages);
  if (!pinned) { std::cerr << ps_last_error() << std::endl; return 1; }
  auto image = [&](int k, int ni, int right) { return pinned + pitch * ((((size_t)ni * S) + k) * 2 + right); };
  for (int k = 0; k < S; k++)
    for (int ni = 0; ni < nImages; ni++)
      for (int right = 0; right < 2; right++) {
        int wi = 0, hi = 0;
        if (!LoadPGM(right ? R[k][ni] : L[k][ni], px, wi, hi) || wi != w || hi != h) { std::cerr << "bad image" << std::endl; return 1; }
        std::memcpy(image(k, ni, right), px.data(), pitch);
      }
  const float fx = (float)calib["Camera.fx"], fy = (float)calib["Camera.fy"], cx = (float)calib["Camera.cx"], cy = (float)calib["Camera.cy"],
              bf = (float)calib["Camera.bf"], thd = (float)calib["ThDepth"];
  try {
    // (a) host-driven
    struct Ref { bool tracked; std::vector<float> tcw; int matches, mapMatches, inliers, state; };
    std::vector<std::vector<Ref>> ref(S);
    {
      ORB_SLAM2::StereoOdometryBatch host(S, fx, fy, cx, cy, bf, w, h, thd);
      std::vector<const uint8_t*> left(S), right(S);
      for (int ni = 0; ni < nImages; ni++) {
        for (int k = 0; k < S; k++) { left[k] = image(k, ni, 0); right[k] = image(k, ni, 1); }
        host.TrackAll(left, right, w);
        for (int k = 0; k < S; k++) {
          ORB_SLAM2::OdoSequence& q = host.sequence(k);
          ref[k].push_back(Ref{q.lastFrameTracked, q.trajectory.back(), q.lastMatches, q.lastMapMatches, q.lastLocalInliers, (int)q.state});
        }
      }
    }
    // (b) device-resident
    ORB_SLAM2::StereoOdometryDevice dev(S, fx, fy, cx, cy, bf, w, h, nImages, thd);
    {
      std::vector<const uint8_t*> left(S), right(S);
      for (int ni = 0; ni < nImages; ni++) {
        for (int k = 0; k < S; k++) { left[k] = image(k, ni, 0); right[k] = image(k, ni, 1); }
        dev.TrackAll(left, right, w);
      }
    }
    std::vector<std::vector<std::vector<float>>> traj;
    std::vector<ps_track_stat> st;
    dev.Fetch(traj, &st);
    int flagDiff = 0, bitDiff = 0, statDiff = 0, tracked = 0, stateDiff = 0;
    double maxAbs = 0;
    for (int k = 0; k < S; k++)
      for (int ni = 0; ni < nImages; ni++) {
        const Ref& r = ref[k][ni];
        const ps_track_stat& d = st[(size_t)ni * S + k];
        if ((d.tracked != 0) != r.tracked) { flagDiff++; continue; }
        if ((int)d.state != r.state) stateDiff++;
        if (!r.tracked) continue;
        tracked++;
        const std::vector<float>& t = traj[k][ni];
        if (std::memcmp(t.data(), r.tcw.data(), 64) != 0) bitDiff++;
        for (int i = 0; i < 16; i++) maxAbs = std::max(maxAbs, (double)std::fabs(t[i] - r.tcw[i]));
        // the counts of Tracking::Track (not defined for the initialisation frame)
        if (ni > 0 && ref[k][ni - 1].state != 0 && d.matches > 0 &&
            (d.matches != r.matches || d.map_matches != r.mapMatches || (d.lm_candidates > 0 && d.lm_inliers != r.inliers))) {
          statDiff++;
          if (statDiff <= 5) std::fprintf(stderr, "seq %d frame %d: device matches %d map %d inliers %d, host %d %d %d\n", k, ni, d.matches, d.map_matches, d.lm_inliers, r.matches, r.mapMatches, r.inliers);
        }
      }
    std::printf("{\"sequences\": %d, \"frames\": %d, \"tracked\": %d, \"tracked_flag_differs\": %d, \"state_differs\": %d, \"pose_bits_differ\": %d, \"max_abs_pose_diff\": %.3g, \"counts_differ\": %d}\n",
                S, nImages, tracked, flagDiff, stateDiff, bitDiff, maxAbs, statDiff);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    ps_pinned_free(pinned);
    return 2;
  }
  ps_pinned_free(pinned);
  return 0;
}
