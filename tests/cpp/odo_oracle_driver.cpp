// Test infrastructure: the C++ host side of the tracking loop (pointslot_amd/host/StereoOdometry.h: OdoSequence) with the CPU
// checker (oracle/liboracle.so) serving its requests instead of the GPU library.  Lets the CPU test suite exercise the C++ state
// machine (request preparation, match application, outlier handling, motion model) against the Python twin of the same loop.
//   usage: odo_oracle_driver path_to_sequence   -> path_to_sequence/CameraTrajectoryOracle.txt
// Also bench.py's camera-chain CPU baseline without Python in the loop: every image is decoded first, the loop is timed on its own
// (last line: "timing: <frames> frames <seconds> s").
#include <chrono>
#include <cstdio>
#include "StereoOdometry.h"
#include "../../examples/kitti_io.h"

extern "C" {
void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void orc_orb_destroy(void* h);
void orc_orb_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* quota, int* umax16);
int orc_orb_run(void* h, const uint8_t* img, int w, int hgt, int stride);
void orc_orb_result(void* h, void* kps28, uint8_t* desc);
int orc_stereo_match(void* hl, void* hr, float mb, float mbf, float* u_right, float* depth);
int orc_search_projection_frame(const ps_proj_train* F, int m, const float* xw, const uint8_t* valid, const int* l_octave, const float* l_angle,
                                const uint8_t* desc, const uint8_t* observed, const float* tcw, const float* tlw, const float* K6,
                                const float* bounds4, const float* scale_factors, float th, int bMono, int check_ori, int* match_of_train);
int orc_search_projection_points(const ps_proj_train* F, int m, const uint8_t* valid, const float* proj_x, const float* proj_y, const float* proj_xr,
                                 const int* level, const float* view_cos, const uint8_t* desc, const uint8_t* observed, const float* scale_factors,
                                 float th, float nnratio, int object, int* match_of_train);
int orc_pose_optimize(int n, const float* xw, const float* obs, const float* inv_sigma2, const uint8_t* valid, float fx, float fy, float cx, float cy,
                      float bf, float* tcw16, uint8_t* outlier, double* trace, int* ntrace);
}

using namespace ORB_SLAM2;

static void serveSearch(ps_proj_problem& p) {
  if (p.frame_mode) {
    const float K6[6] = {p.fx, p.fy, p.cx, p.cy, p.mbf, p.mb};
    p.nmatches = orc_search_projection_frame(&p.train, p.nq, p.q_xw, p.q_valid, p.q_octave, p.q_angle, p.q_desc, p.q_observed, p.tcw, p.tlw, K6, p.bounds,
                                             p.scale_factors, p.th, p.mono, p.check_orientation, p.match_of_train);
  } else {
    // the checker takes the predicted level and the viewing cosine and derives the window itself: recover them from the request
    std::vector<int> level(p.nq);
    std::vector<float> viewCos(p.nq);
    for (int i = 0; i < p.nq; i++) {
      level[i] = p.q_max_level[i];
      const float r = p.q_valid[i] ? p.q_radius[i] / p.scale_factors[level[i]] : 4.f;
      viewCos[i] = r < 3.f ? 0.999f : 0.9f;   // ORBmatcher::RadiusByViewingCos: 2.5 above 0.998, else 4.0
    }
    p.nmatches = orc_search_projection_points(&p.train, p.nq, p.q_valid, p.q_u, p.q_v, p.q_ur, level.data(), viewCos.data(), p.q_desc, p.q_observed,
                                              p.scale_factors, p.th, p.nn_ratio, p.use_bbox, p.match_of_train);
  }
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "Usage: odo_oracle_driver path_to_sequence\n"); return 1; }
  const std::string seq = argv[1];
  std::vector<std::string> vl, vr;
  std::vector<double> ts;
  LoadImages(seq, vl, vr, ts);
  if (vl.empty()) { std::fprintf(stderr, "no images under %s\n", seq.c_str()); return 1; }
  std::map<std::string, double> calib = LoadCalib(seq);
  void* exl = orc_orb_create(2000, 1.2f, 8, 20, 5);
  void* exr = orc_orb_create(2000, 1.2f, 8, 20, 5);
  std::vector<float> sf(8), isf(8), s2(8), is2(8);
  std::vector<int> quota(8), umax(16);
  orc_orb_tables(exl, sf.data(), isf.data(), s2.data(), is2.data(), quota.data(), umax.data());
  int w = 0, h = 0;
  {
    std::vector<unsigned char> first;
    if (!LoadPGM(vl[0], first, w, h)) return 1;
  }
  OdoCamera cam((float)calib["Camera.fx"], (float)calib["Camera.fy"], (float)calib["Camera.cx"], (float)calib["Camera.cy"], (float)calib["Camera.bf"], w, h,
                (float)calib["ThDepth"], sf, is2);
  OdoSequence odo(&cam);
  std::vector<std::vector<unsigned char>> allL(vl.size()), allR(vl.size());
  for (size_t ni = 0; ni < vl.size(); ni++) {
    int wr, hr;
    if (!LoadPGM(vl[ni], allL[ni], w, h) || !LoadPGM(vr[ni], allR[ni], wr, hr)) return 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t ni = 0; ni < vl.size(); ni++) {
    const std::vector<unsigned char>&L = allL[ni], &R = allR[ni];
    std::unique_ptr<OdoFrame> F(new OdoFrame);
    const int n = orc_orb_run(exl, L.data(), w, h, w);
    orc_orb_run(exr, R.data(), w, h, w);
    F->mvKeys.resize(n); F->mDescriptors.create(std::max(n, 1), 32, 0); F->mDescriptors.rows = n;
    orc_orb_result(exl, F->mvKeys.data(), F->mDescriptors.data);
    F->mvuRight.assign(n, -1.f); F->mvDepth.assign(n, -1.f);
    orc_stereo_match(exl, exr, cam.mb, cam.mbf, F->mvuRight.data(), F->mvDepth.data());
    OdoSequence::Request rq = odo.begin(std::move(F));
    while (rq != OdoSequence::NONE) {
      if (rq == OdoSequence::SEARCH) serveSearch(odo.proj);
      else {
        ps_pose_problem& p = odo.posep;
        p.result = orc_pose_optimize(p.n, p.xw, p.obs, p.inv_sigma2, p.valid, p.fx, p.fy, p.cx, p.cy, p.bf, p.tcw, p.outlier, nullptr, nullptr);
      }
      rq = odo.advance();
    }
    std::printf("frame %zu: %s matches %d map %d local inliers %d\n", ni, odo.lastFrameTracked ? "ok" : "not tracked", odo.lastMatches, odo.lastMapMatches,
                odo.lastLocalInliers);
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  SaveTrajectoryKITTI(seq + "/CameraTrajectoryOracle.txt", odo.trajectory);
  std::printf("timing: %zu frames %.6f s\n", vl.size(), secs);
  orc_orb_destroy(exl); orc_orb_destroy(exr);
  return 0;
}
