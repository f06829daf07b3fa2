// g2o::ObjectState (/root/reference/include/g2o_Object.h:30-93, src/g2o_Object.cc:58-182) and the
// declarations of the object edges the reference defines but never instantiates (row a21), as host code without Eigen:
// the callers of the object optimisers (Optimizer.cc:510,553: GetCFInFrameObjState(...).pose) keep object poses in this
// type.  Nothing here runs on the device: the hot path takes poses as 7 doubles (SE3Quat::toVector()).
//
// A minimal Eigen-shaped layer (operator() element access, fixed sizes) keeps reference-shaped code compiling; a maintainer
// who builds against the real g2o / Eigen keeps the reference's own header instead of this one.
#pragma once
#include <algorithm>
#include <cmath>
#include "../csrc/se3.h"

namespace g2o {

template <int N> struct VecN {
  double v[N];
  VecN() { for (int i = 0; i < N; i++) v[i] = 0; }
  double& operator()(int i) { return v[i]; }
  const double& operator()(int i) const { return v[i]; }
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
  void setZero() { for (int i = 0; i < N; i++) v[i] = 0; }
};
typedef VecN<2> Vector2d;
typedef VecN<3> Vector3d;
typedef VecN<4> Vector4d;
typedef VecN<6> Vector6d;
typedef VecN<9> Vector9d;
inline Vector3d Vec3(double x, double y, double z) { Vector3d r; r(0) = x; r(1) = y; r(2) = z; return r; }
template <int R, int C> struct MatRC {
  double m[R * C];   // row-major
  MatRC() { for (int i = 0; i < R * C; i++) m[i] = 0; }
  double& operator()(int r, int c) { return m[r * C + c]; }
  const double& operator()(int r, int c) const { return m[r * C + c]; }
};
typedef MatRC<3, 3> Matrix3d;
typedef MatRC<4, 4> Matrix4d;
typedef MatRC<3, 8> Matrix3x8d;
struct Quaterniond { double x = 0, y = 0, z = 0, w = 1; };

// matrix_utils.cc:18-31
inline Quaterniond zyx_euler_to_quat(const double& roll, const double& pitch, const double& yaw) {
  const double sy = std::sin(yaw * 0.5), cy = std::cos(yaw * 0.5), sp = std::sin(pitch * 0.5), cp = std::cos(pitch * 0.5);
  const double sr = std::sin(roll * 0.5), cr = std::cos(roll * 0.5);
  Quaterniond q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  return q;
}

// the members of g2o::SE3Quat (Thirdparty/g2o/g2o/types/se3quat.h) that ObjectState and its callers use, on the POD of se3.h
class SE3Quat {
 public:
  SE3Quat() { T.q[0] = T.q[1] = T.q[2] = 0; T.q[3] = 1; T.t[0] = T.t[1] = T.t[2] = 0; }
  explicit SE3Quat(const Se3& t) : T(t) {}
  SE3Quat(const Quaterniond& q, const Vector3d& t) {   // se3quat.h:62-69: the rotation is normalised (w >= 0)
    T.q[0] = q.x; T.q[1] = q.y; T.q[2] = q.z; T.q[3] = q.w;
    for (int i = 0; i < 3; i++) T.t[i] = t(i);
    se3_normalize(T);
  }
  static SE3Quat fromVector(const double* p7) {        // (tx, ty, tz, qx, qy, qz, qw) = toVector()
    SE3Quat r;
    for (int i = 0; i < 3; i++) r.T.t[i] = p7[i];
    for (int i = 0; i < 4; i++) r.T.q[i] = p7[3 + i];
    return r;
  }
  void toVector(double* p7) const { for (int i = 0; i < 3; i++) p7[i] = T.t[i]; for (int i = 0; i < 4; i++) p7[3 + i] = T.q[i]; }
  Vector3d translation() const { return Vec3(T.t[0], T.t[1], T.t[2]); }
  Quaterniond rotation() const { Quaterniond q; q.x = T.q[0]; q.y = T.q[1]; q.z = T.q[2]; q.w = T.q[3]; return q; }
  void setTranslation(const Vector3d& t) { for (int i = 0; i < 3; i++) T.t[i] = t(i); }
  void setRotation(const Quaterniond& q) { T.q[0] = q.x; T.q[1] = q.y; T.q[2] = q.z; T.q[3] = q.w; }
  void normalizeRotation() { se3_normalize(T); }
  SE3Quat operator*(const SE3Quat& o) const { return SE3Quat(se3_mul(T, o.T)); }
  Vector3d operator*(const Vector3d& x) const { double o[3]; se3_map(T, x.v, o); return Vec3(o[0], o[1], o[2]); }   // map()
  SE3Quat inverse() const {                                                  // se3quat.h:112-117
    Se3 r;
    r.q[0] = -T.q[0]; r.q[1] = -T.q[1]; r.q[2] = -T.q[2]; r.q[3] = T.q[3];
    const double nt[3] = {T.t[0] * -1., T.t[1] * -1., T.t[2] * -1.};
    se3_rotate(r.q, nt, r.t);
    return SE3Quat(r);
  }
  static SE3Quat exp(const Vector6d& u) { return SE3Quat(se3_exp(u.v, false)); }
  Matrix4d to_homogeneous_matrix() const {                                   // se3quat.h:296-304
    double R[9];
    se3_quat_to_R(T.q, R);
    Matrix4d M;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M(r, c) = R[3 * r + c]; M(r, 3) = T.t[r]; }
    M(3, 3) = 1;
    return M;
  }
  Matrix3d rotationMatrix() const { double R[9]; se3_quat_to_R(T.q, R); Matrix3d M; for (int i = 0; i < 9; i++) M.m[i] = R[i]; return M; }
  Se3 T;
};

// ORB_SLAM2::EnObjectCenter (src/Parameters.cc:60): 0 = object frame at the geometric centre, 1 = at the bottom centre
// (virtual KITTI).  The reference reads it from "Viewer.ObjectCenter"; here it is a parameter of the calls that need it.
class ObjectState {
 public:
  SE3Quat pose;
  Vector3d scale;   // the full extents (length, height, width along x, y, z), not half of them
  ObjectState() {}
  ObjectState(const SE3Quat& se3Pose, const Vector3d& eigScale) : pose(se3Pose), scale(eigScale) {}
  // xyz roll pitch yaw scale
  inline void fromMinimalVector(const Vector9d& v) {
    pose = SE3Quat(zyx_euler_to_quat(v(3), v(4), v(5)), Vec3(v(0), v(1), v(2)));
    scale = Vec3(v(6), v(7), v(8));
  }
  inline Vector3d translation() const { return pose.translation(); }
  inline void setPose(const SE3Quat& se3Pose) { pose = se3Pose; }
  inline void setScale(const Vector3d& scale_) { scale = scale_; }
  inline void setTranslation(const Vector3d& t_) { pose.setTranslation(t_); }
  inline void setRotation(const Quaterniond& r_) { pose.setRotation(r_); }
  inline void setRotation(const Vector3d& v) { pose.setRotation(zyx_euler_to_quat(v(0), v(1), v(2))); pose.normalizeRotation(); }

  ObjectState transform_from(const SE3Quat& Twc) const { return ObjectState(Twc * pose, scale); }   // g2o_Object.cc:81-87

  // g2o_Object.cc:58-79: pose <- pose * [exp(omega * dt) | v * dt]; LastVel = (omega, v)
  void UsingVelocitySetPredictPos(const Vector6d& LastVel, const double& delta_t) {
    Vector6d delta_pos;
    for (int i = 0; i < 3; i++) delta_pos(i) = LastVel(i) * delta_t;
    SE3Quat Tlc = SE3Quat::exp(delta_pos);
    Tlc.setTranslation(Vec3(LastVel(3) * delta_t, LastVel(4) * delta_t, LastVel(5) * delta_t));
    pose = pose * Tlc;
  }

  // [R * diag(scale / 2) | t; 0 1]  (g2o_Object.cc:91-98)
  Matrix4d similarityTransform() const {
    Matrix4d res = pose.to_homogeneous_matrix();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) res(r, c) = res(r, c) * (scale(c) * 0.5);
    return res;
  }

  // the 8 corners in the frame `pose` maps into (g2o_Object.cc:101-137)
  Matrix3x8d compute3D_BoxCorner(int objectCenter = 0) const {
    static const double centre[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
    static const double bottom[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {0, 0, 0, 0, -2, -2, -2, -2}, {1, -1, -1, 1, 1, -1, -1, 1}};
    const double (*body)[8] = objectCenter == 0 ? centre : bottom;
    const Matrix4d S = similarityTransform();
    Matrix3x8d out;
    for (int k = 0; k < 8; k++) {
      double h[4];
      for (int r = 0; r < 4; r++) h[r] = S(r, 0) * body[0][k] + S(r, 1) * body[1][k] + S(r, 2) * body[2][k] + S(r, 3) * 1.0;
      for (int r = 0; r < 3; r++) out(r, k) = h[r] / h[3];
    }
    return out;
  }

  // [u_min v_min u_max v_max] of the 8 projected corners (g2o_Object.cc:140-153)
  Vector4d projectOntoImageRect(const SE3Quat& campose_cw, const Matrix3d& Kalib, int objectCenter = 0) const {
    return rect(compute3D_BoxCorner(objectCenter), &campose_cw, Kalib);
  }
  // the same when `pose` already maps into the camera frame (g2o_Object.cc:156-169)
  Vector4d projectOntoImageRectFromCamera(const Matrix3d& Kalib, int objectCenter = 0) const {
    return rect(compute3D_BoxCorner(objectCenter), nullptr, Kalib);
  }
  // [center.x center.y width height] (g2o_Object.cc:172-182)
  Vector4d projectOntoImageBbox(const SE3Quat& campose_cw, const Matrix3d& Kalib, int objectCenter = 0) const {
    const Vector4d r = projectOntoImageRect(campose_cw, Kalib, objectCenter);
    Vector4d o;
    o(0) = (r(2) + r(0)) / 2; o(1) = (r(3) + r(1)) / 2; o(2) = r(2) - r(0); o(3) = r(3) - r(1);
    return o;
  }

 private:
  static Vector4d rect(const Matrix3x8d& corners, const SE3Quat* Tcw, const Matrix3d& K) {
    Matrix4d M;
    if (Tcw) M = Tcw->to_homogeneous_matrix();
    double lo[2] = {0, 0}, hi[2] = {0, 0};
    for (int k = 0; k < 8; k++) {
      double c[3] = {corners(0, k), corners(1, k), corners(2, k)};
      if (Tcw) {
        double h[4];
        for (int r = 0; r < 4; r++) h[r] = M(r, 0) * c[0] + M(r, 1) * c[1] + M(r, 2) * c[2] + M(r, 3) * 1.0;
        for (int r = 0; r < 3; r++) c[r] = h[r] / h[3];
      }
      double p[3];
      for (int r = 0; r < 3; r++) p[r] = K(r, 0) * c[0] + K(r, 1) * c[1] + K(r, 2) * c[2];
      const double uv[2] = {p[0] / p[2], p[1] / p[2]};
      for (int r = 0; r < 2; r++) { lo[r] = k == 0 ? uv[r] : std::min(lo[r], uv[r]); hi[r] = k == 0 ? uv[r] : std::max(hi[r], uv[r]); }
    }
    Vector4d o;
    o(0) = lo[0]; o(1) = lo[1]; o(2) = hi[0]; o(3) = hi[1];
    return o;
  }
};

// ---- a21: an object edge the reference defines and never instantiates (g2o_Object.cc:404-480), kept for API
// compatibility.  Vertices: a cuboid (object-to-world pose) and a point in the object frame; Tcw is a constant of the edge.
// With Tcw = I the residual and both Jacobians are those of EdgeStereoSE3ProjectXYZ on (Tco, point) - the edge of
// Optimizer::ObjectLocalBundleAdjustment that the ba_* kernels implement (tests/test_object_state.py checks the reduction). ----
struct EdgeStereoDynamicPointAndCuboid {
  SE3Quat Tcw;
  Matrix3d Kalib;
  double bf = 0;
  Vector3d _measurement, _error;
  double _jacobianOplusXi[3][6], _jacobianOplusXj[3][3];
  bool whether_fixrotation = false;

  Vector3d cam_project(const Vector3d& trans_xyz) const {   // :418-425 (float invz, as the reference)
    const float invz = 1.0f / trans_xyz[2];
    Vector3d res;
    res[0] = trans_xyz[0] * invz * Kalib(0, 0) + Kalib(0, 2);
    res[1] = trans_xyz[1] * invz * Kalib(1, 1) + Kalib(1, 2);
    res[2] = res[0] - bf * invz;
    return res;
  }
  void computeError(const ObjectState& cuboid, const Vector3d& point) {   // :404-411
    const Vector3d localpt = Tcw * (cuboid.pose * point);
    const Vector3d proj = cam_project(localpt);
    for (int i = 0; i < 3; i++) _error[i] = _measurement[i] - proj[i];
  }
  void linearizeOplus(const ObjectState& cuboid, const Vector3d& objectpt) {   // :427-480
    const SE3Quat combinedT = Tcw * cuboid.pose;
    const Vector3d camerapt = combinedT * objectpt;
    const double fx = Kalib(0, 0), fy = Kalib(1, 1);
    const double x = camerapt[0], y = camerapt[1], z = camerapt[2], z_2 = z * z;
    const double P[3][3] = {{fx / z, 0, -x * fx / z_2}, {0, fy / z, -y * fy / z_2}, {fx / z, 0, (-fx * x + bf) / z_2}};
    const Vector3d Pwf = cuboid.pose * objectpt;
    const Matrix3d R = Tcw.rotationMatrix();
    const double S[3][3] = {{0, -Pwf[2], Pwf[1]}, {Pwf[2], 0, -Pwf[0]}, {-Pwf[1], Pwf[0], 0}};   // skew(Pwf)
    double temp[3][6];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        temp[r][c] = -(R(r, 0) * S[0][c] + R(r, 1) * S[1][c] + R(r, 2) * S[2][c]);
        temp[r][3 + c] = R(r, c);
      }
    const Matrix3d Rc = combinedT.rotationMatrix();
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 6; c++) _jacobianOplusXi[r][c] = -(P[r][0] * temp[0][c] + P[r][1] * temp[1][c] + P[r][2] * temp[2][c]);
      for (int c = 0; c < 3; c++) _jacobianOplusXj[r][c] = -(P[r][0] * Rc(0, c) + P[r][1] * Rc(1, c) + P[r][2] * Rc(2, c));
    }
    if (whether_fixrotation)
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) _jacobianOplusXi[r][c] = 0;
  }
};

}  // namespace g2o
