// Usage: stereo_kitti_batch [--max-frames N] [--device D] [--groups G] path_to_sequence_1 path_to_sequence_2 ...
//
// BASELINE config 4 on one GPU: several independent stereo sequences (the reference's on-disk layout, see kitti_io.h) tracked
// in lockstep by ORB_SLAM2::StereoOdometryBatch — every step extracts all sequences' stereo pairs as one batch and serves
// each round of SearchByProjection / PoseOptimization problems with one C-ABI call.  --groups G splits the sequences over G
// such batches on G threads (own handles), which overlap one group's transfers with another's kernels.  All images are loaded into page-locked
// memory before the clock starts (the figure is tracking throughput, not disk or PGM decoding).  Writes
// CameraTrajectoryBatch.txt next to each sequence (System::SaveTrajectoryKITTI format) and prints one JSON line.
//   g++ -std=c++17 -O2 -pthread -I pointslot_amd/host -I include examples/stereo_kitti_batch.cpp -L pointslot_amd -lpointslot_hip
#include <chrono>
#include <cstdio>
#include "StereoOdometry.h"
#include "kitti_io.h"

int main(int argc, char** argv) {
  std::vector<std::string> seqs;
  int maxFrames = 1 << 30, device = 0, groups = 1;
  for (int a = 1; a < argc; a++) {
    if (std::string(argv[a]) == "--max-frames" && a + 1 < argc) maxFrames = std::atoi(argv[++a]);
    else if (std::string(argv[a]) == "--device" && a + 1 < argc) device = std::atoi(argv[++a]);
    else if (std::string(argv[a]) == "--groups" && a + 1 < argc) groups = std::atoi(argv[++a]);
    else seqs.push_back(argv[a]);
  }
  if (seqs.empty()) { std::cerr << "Usage: ./stereo_kitti_batch [--max-frames N] [--device D] [--groups G] path_to_sequence ..." << std::endl; return 1; }
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
  // layout: step-major, inside a step group by group (sequence k belongs to group k % G), left / right interleaved - so the
  // images of one group's step are one contiguous block in the order the batch wants them and go up in a single transfer
  const int G = std::max(1, std::min(groups, S));
  std::vector<int> groupBase(G + 1, 0);
  for (int g = 0; g < G; g++) groupBase[g + 1] = groupBase[g] + (S - g + G - 1) / G;
  auto image = [&](int k, int ni, int right) { return pinned + pitch * ((((size_t)ni * S) + groupBase[k % G] + k / G) * 2 + right); };
  for (int k = 0; k < S; k++)
    for (int ni = 0; ni < nImages; ni++)
      for (int right = 0; right < 2; right++) {
        int wi = 0, hi = 0;
        const std::string& path = right ? vstrRight[k][ni] : vstrLeft[k][ni];
        if (!LoadPGM(path, px, wi, hi) || wi != w || hi != h) { std::cerr << "Failed to load image at: " << path << std::endl; return 1; }
        std::memcpy(image(k, ni, right), px.data(), pitch);
      }
  // `groups` StereoOdometryBatch instances, each with its own handles and its own thread, share the sequences round-robin:
  // while one group waits for a transfer or a kernel the others keep the GPU and the PCIe link busy
  const bool prefetch = std::getenv("PS_ODO_NO_PREFETCH") == nullptr;   // queue step n+1's extraction during step n's search / pose rounds
  const int warm = nImages > 2 ? 2 : (nImages > 1 ? 1 : 0);             // step 0 initialises, step 1 sizes the staging buffers
  struct Group {
    std::vector<int> members;
    std::vector<double> times, parts[4];
    int lost = 0, rounds = 0;
    std::chrono::steady_clock::time_point tStart, tEnd;
    double unixStart = 0, unixEnd = 0;   // the same instants on the system clock: several driver processes can be laid side by side
    std::string error;
  };
  std::vector<Group> grp(G);
  for (int k = 0; k < S; k++) grp[k % G].members.push_back(k);
  std::vector<std::vector<std::vector<float>>> trajectories(S);
  std::mutex mtx;
  std::condition_variable cv;
  int arrived = 0;
  std::cout << "Start processing " << S << " sequences in " << G << " lockstep group(s) ... Images per sequence: " << nImages << std::endl;
  auto runGroup = [&](int g) {
    Group& R = grp[g];
    const int n = (int)R.members.size();
    try {
      ORB_SLAM2::StereoOdometryBatch SLAM(n, (float)calib["Camera.fx"], (float)calib["Camera.fy"], (float)calib["Camera.cx"], (float)calib["Camera.cy"],
                                          (float)calib["Camera.bf"], w, h, (float)calib["ThDepth"], 2000, 1.2f, 8, 20, 5, device);
      R.times.assign(nImages, 0.0);
      for (auto& v : R.parts) v.assign(nImages, 0.0);
      std::vector<const uint8_t*> left(n), right(n), nextLeft(n), nextRight(n);
      for (int ni = 0; ni < nImages; ni++) {
        if (ni == warm) {   // all groups start the timed steps together
          std::unique_lock<std::mutex> lk(mtx);
          if (++arrived == G) cv.notify_all(); else cv.wait(lk, [&]() { return arrived >= G; });
          R.tStart = std::chrono::steady_clock::now();
          R.unixStart = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        }
        // every timed step pays for its own extraction: the last warm-up step queues nothing ahead (the extraction of step
        // `warm` then starts inside the timed window), and the last step has nothing to queue
        const bool more = prefetch && ni + 1 < nImages && ni + 1 != warm;
        for (int i = 0; i < n; i++) {
          const int k = R.members[i];
          left[i] = image(k, ni, 0); right[i] = image(k, ni, 1);
          if (more) { nextLeft[i] = image(k, ni + 1, 0); nextRight[i] = image(k, ni + 1, 1); }
        }
        const double e0 = SLAM.tExtract, h0 = SLAM.tHost, s0 = SLAM.tSearch, p0 = SLAM.tPose;
        const auto t1 = std::chrono::steady_clock::now();
        const int tracked = SLAM.TrackAll(left, right, w, more ? &nextLeft : nullptr, more ? &nextRight : nullptr);
        const auto t2 = std::chrono::steady_clock::now();
        R.times[ni] = std::chrono::duration<double>(t2 - t1).count();
        R.lost += n - tracked;
        R.parts[0][ni] = SLAM.tExtract - e0; R.parts[1][ni] = SLAM.tHost - h0; R.parts[2][ni] = SLAM.tSearch - s0; R.parts[3][ni] = SLAM.tPose - p0;
        if (G == 1)
          std::printf("step %d: %d of %d sequences tracked, %.3f ms (frames %.3f, host %.3f, search %.3f, pose %.3f)\n", ni, tracked, n, 1e3 * R.times[ni],
                      1e3 * R.parts[0][ni], 1e3 * R.parts[1][ni], 1e3 * R.parts[2][ni], 1e3 * R.parts[3][ni]);
      }
      R.tEnd = std::chrono::steady_clock::now();
      R.unixEnd = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
      R.rounds = SLAM.rounds;
      for (int i = 0; i < n; i++) trajectories[R.members[i]] = SLAM.sequence(i).trajectory;
    } catch (const std::exception& e) {
      R.error = e.what();
      std::unique_lock<std::mutex> lk(mtx);   // do not leave the other groups waiting at the start line
      if (arrived < G) { arrived = G; cv.notify_all(); }
    }
  };
  {
    std::vector<std::thread> threads;
    for (int g = 1; g < G; g++) threads.emplace_back(runGroup, g);
    runGroup(0);
    for (std::thread& t : threads) t.join();
  }
  for (const Group& R : grp)
    if (!R.error.empty()) { std::cerr << "error: " << R.error << std::endl; ps_pinned_free(pinned); return 2; }
  for (int k = 0; k < S; k++) SaveTrajectoryKITTI(seqs[k] + "/CameraTrajectoryBatch.txt", trajectories[k]);
  // statistics over the timed steps: wall clock from the common start to the last group's end
  auto medianOf = [&](const std::vector<double>& v) {
    std::vector<double> t(v.begin() + warm, v.end());
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  double median = 0, wall = 0, unixStart = 1e300, unixEnd = 0;
  int lost = 0;
  for (const Group& R : grp) {
    unixStart = std::min(unixStart, R.unixStart); unixEnd = std::max(unixEnd, R.unixEnd);
    median = std::max(median, medianOf(R.times));
    wall = std::max(wall, std::chrono::duration<double>(R.tEnd - grp[0].tStart).count());
    lost += R.lost;
  }
  const int timedSteps = nImages - warm;
  std::printf("{\"sequences\": %d, \"groups\": %d, \"frames_per_sequence\": %d, \"timed_steps\": %d, \"untracked_frames\": %d, \"median_ms_per_step\": %.4f, "
              "\"wall_ms_timed_steps\": %.4f, \"frames_per_s\": %.1f, \"ms_per_step_frames\": %.4f, \"ms_per_step_host\": %.4f, \"ms_per_step_search\": %.4f, "
              "\"ms_per_step_pose\": %.4f, \"device_rounds_per_step\": %.2f, \"unix_start\": %.6f, \"unix_end\": %.6f}\n",
              S, G, nImages, timedSteps, lost, 1e3 * median, 1e3 * wall, timedSteps > 0 ? S * timedSteps / wall : 0.0, 1e3 * medianOf(grp[0].parts[0]),
              1e3 * medianOf(grp[0].parts[1]), 1e3 * medianOf(grp[0].parts[2]), 1e3 * medianOf(grp[0].parts[3]), grp[0].rounds / (double)nImages, unixStart, unixEnd);
  ps_pinned_free(pinned);
  return 0;
}
