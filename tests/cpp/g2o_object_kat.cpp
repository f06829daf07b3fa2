// Known-answer driver for pointslot_amd/host/g2o_Object.h (g2o::ObjectState): reads one command per
// line from stdin, prints the results as numbers; tests/test_object_state.py compares them with independent numpy / scipy
// computations and, for the a21 edge, with the CPU checker's a16 stereo edge.  Host code only, no GPU.
//   state    p7 scale3 centre cam7 fx fy cx cy      -> corners(24, row-major 3x8) rect(4) bbox(4) rect_from_camera(4)
//   predict  p7 vel6 dt                             -> p7
//   minimal  v9                                     -> p7 scale3
//   edge     tco7 point3 obs3 fx fy cx cy bf tcw7   -> err3 Ji(18) Jj(9)
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include "g2o_Object.h"

static g2o::SE3Quat read7(std::istream& in) { double p[7]; for (double& v : p) in >> v; return g2o::SE3Quat::fromVector(p); }
static void print7(const g2o::SE3Quat& T) { double p[7]; T.toVector(p); for (double v : p) std::printf("%.17g ", v); }

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string cmd;
    if (!(in >> cmd)) continue;
    if (cmd == "state") {
      g2o::ObjectState st;
      st.pose = read7(in);
      for (int i = 0; i < 3; i++) in >> st.scale(i);
      int centre; in >> centre;
      const g2o::SE3Quat cam = read7(in);
      g2o::Matrix3d K; in >> K(0, 0) >> K(1, 1) >> K(0, 2) >> K(1, 2); K(2, 2) = 1;
      const g2o::Matrix3x8d c = st.compute3D_BoxCorner(centre);
      for (int i = 0; i < 24; i++) std::printf("%.17g ", c.m[i]);
      const g2o::Vector4d r = st.projectOntoImageRect(cam, K, centre), b = st.projectOntoImageBbox(cam, K, centre);
      const g2o::Vector4d rc = st.transform_from(cam).projectOntoImageRectFromCamera(K, centre);
      for (int i = 0; i < 4; i++) std::printf("%.17g ", r(i));
      for (int i = 0; i < 4; i++) std::printf("%.17g ", b(i));
      for (int i = 0; i < 4; i++) std::printf("%.17g ", rc(i));
    } else if (cmd == "predict") {
      g2o::ObjectState st;
      st.pose = read7(in);
      g2o::Vector6d vel; for (int i = 0; i < 6; i++) in >> vel(i);
      double dt; in >> dt;
      st.UsingVelocitySetPredictPos(vel, dt);
      print7(st.pose);
    } else if (cmd == "minimal") {
      g2o::Vector9d v; for (int i = 0; i < 9; i++) in >> v(i);
      g2o::ObjectState st;
      st.fromMinimalVector(v);
      print7(st.pose);
      for (int i = 0; i < 3; i++) std::printf("%.17g ", st.scale(i));
    } else if (cmd == "edge") {
      g2o::ObjectState st;
      st.pose = read7(in);
      g2o::Vector3d pt, obs;
      for (int i = 0; i < 3; i++) in >> pt(i);
      for (int i = 0; i < 3; i++) in >> obs(i);
      g2o::EdgeStereoDynamicPointAndCuboid e;
      in >> e.Kalib(0, 0) >> e.Kalib(1, 1) >> e.Kalib(0, 2) >> e.Kalib(1, 2) >> e.bf;
      e.Kalib(2, 2) = 1;
      e.Tcw = read7(in);
      e._measurement = obs;
      e.computeError(st, pt);
      e.linearizeOplus(st, pt);
      for (int i = 0; i < 3; i++) std::printf("%.17g ", e._error[i]);
      for (int r = 0; r < 3; r++) for (int c = 0; c < 6; c++) std::printf("%.17g ", e._jacobianOplusXi[r][c]);
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) std::printf("%.17g ", e._jacobianOplusXj[r][c]);
    }
    std::printf("\n");
  }
  return 0;
}
