// SE3 algebra shared by the optimiser kernels and their host marshalling: g2o::SE3Quat semantics
// (/root/reference/Thirdparty/g2o/g2o/types/se3quat.h) on a POD {quaternion (x,y,z,w), translation}.
// FP64 throughout.  Every function states the reference lines it follows.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define PS_HD __host__ __device__ __forceinline__
#else
#define PS_HD inline
#endif

struct Se3 { double q[4]; double t[3]; };

// Eigen::Quaterniond(Matrix3d) as used by SE3Quat(R, t) (se3quat.h:58): Shepperd's method
PS_HD void se3_quat_from_R(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
  } else if (R[0] >= R[4] && R[0] >= R[8]) {
    double s = sqrt(R[0] - R[4] - R[8] + 1.0);
    q[0] = 0.5 * s; s = 0.5 / s;
    q[3] = (R[7] - R[5]) * s; q[1] = (R[3] + R[1]) * s; q[2] = (R[6] + R[2]) * s;
  } else if (R[4] > R[0] && R[4] >= R[8]) {
    double s = sqrt(R[4] - R[8] - R[0] + 1.0);
    q[1] = 0.5 * s; s = 0.5 / s;
    q[3] = (R[2] - R[6]) * s; q[2] = (R[7] + R[5]) * s; q[0] = (R[1] + R[3]) * s;
  } else {
    double s = sqrt(R[8] - R[0] - R[4] + 1.0);
    q[2] = 0.5 * s; s = 0.5 / s;
    q[3] = (R[3] - R[1]) * s; q[0] = (R[2] + R[6]) * s; q[1] = (R[5] + R[7]) * s;
  }
}
// SE3Quat::normalizeRotation (se3quat.h:306-311): w >= 0, unit norm
PS_HD void se3_normalize(Se3& T) {
  if (T.q[3] < 0) { T.q[0] = -T.q[0]; T.q[1] = -T.q[1]; T.q[2] = -T.q[2]; T.q[3] = -T.q[3]; }
  const double n = sqrt(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
  T.q[0] /= n; T.q[1] /= n; T.q[2] /= n; T.q[3] /= n;
}
// Eigen toRotationMatrix
PS_HD void se3_quat_to_R(const double q[4], double R[9]) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Quaterniond * Vector3d (Eigen _transformVector)
PS_HD void se3_rotate(const double q[4], const double v[3], double o[3]) {
  double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
  ux += ux; uy += uy; uz += uz;
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
// SE3Quat::map (se3quat.h:242-245)
PS_HD void se3_map(const Se3& T, const double x[3], double o[3]) {
  se3_rotate(T.q, x, o);
  o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2];
}
// SE3Quat::operator* (se3quat.h:104-110)
PS_HD Se3 se3_mul(const Se3& a, const Se3& b) {
  Se3 r;
  double rt[3];
  se3_rotate(a.q, b.t, rt);
  r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
  r.q[3] = a.q[3] * b.q[3] - a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2];
  r.q[0] = a.q[3] * b.q[0] + a.q[0] * b.q[3] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
  r.q[1] = a.q[3] * b.q[1] + a.q[1] * b.q[3] + a.q[2] * b.q[0] - a.q[0] * b.q[2];
  r.q[2] = a.q[3] * b.q[2] + a.q[2] * b.q[3] + a.q[0] * b.q[1] - a.q[1] * b.q[0];
  se3_normalize(r);
  return r;
}
// SE3Quat::inverse (se3quat.h:112-117): conjugate rotation, t = q^-1 * (t * -1)
PS_HD Se3 se3_inverse(const Se3& a) {
  Se3 r;
  r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
  const double nt[3] = {a.t[0] * -1., a.t[1] * -1., a.t[2] * -1.};
  se3_rotate(r.q, nt, r.t);
  return r;
}
PS_HD Se3 se3_from_Rt(const double R[9], const double t[3]) {
  Se3 T;
  se3_quat_from_R(R, T.q);
  T.t[0] = t[0]; T.t[1] = t[1]; T.t[2] = t[2];
  se3_normalize(T);
  return T;
}
// SE3Quat::exp (se3quat.h:248-279).  norollpitch: exptwist_norollpitch (src/g2o_Object.cc:26-56) —
// R = Rz(omega_z), V from the full Rodrigues series of omega.
// sin / cos / cube for the exponential map.  On the device the library routines (argument reduction for any magnitude, a generic
// pow) cost more than the rest of the 6x6 solve they sit in: angles below pi/4 - every LM step in practice - take the classic
// polynomial kernels directly and x^3 is two multiplications (<= 1 ulp from the library values; the optimisers are a tolerance
// target).  The host build keeps libm.
#if defined(__HIP_DEVICE_COMPILE__)
PS_HD void se3_sincos(double x, double& sn, double& cs) {
  if (fabs(x) < 0.78539816339744830962) {
    const double z = x * x;
    double ps = 1.58962301576546568060e-10;
    ps = fma(ps, z, -2.50507477628578072866e-8); ps = fma(ps, z, 2.75573136213857245213e-6); ps = fma(ps, z, -1.98412698295895385996e-4);
    ps = fma(ps, z, 8.33333333332211858878e-3); ps = fma(ps, z, -1.66666666666666307295e-1);
    sn = fma(x * z, ps, x);
    double pc = -1.13585365213876817300e-11;
    pc = fma(pc, z, 2.08757008419747316778e-9); pc = fma(pc, z, -2.75573141792967388112e-7); pc = fma(pc, z, 2.48015872888517045348e-5);
    pc = fma(pc, z, -1.38888888888730564116e-3); pc = fma(pc, z, 4.16666666666665929218e-2);
    cs = fma(z * z, pc, fma(-0.5, z, 1.0));
  } else { sn = sin(x); cs = cos(x); }
}
PS_HD double se3_cube(double x) { return x * x * x; }
#else
PS_HD void se3_sincos(double x, double& sn, double& cs) { sn = sin(x); cs = cos(x); }
PS_HD double se3_cube(double x) { return pow(x, 3); }
#endif
PS_HD Se3 se3_exp(const double u[6], bool norollpitch) {
  const double wx = u[0], wy = u[1], wz = u[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double Om[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double Om2[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) Om2[r * 3 + c] = Om[r * 3] * Om[c] + Om[r * 3 + 1] * Om[3 + c] + Om[r * 3 + 2] * Om[6 + c];
  double R[9], V[9];
  if (norollpitch) {
    double c, s;
    se3_sincos(wz, s, c);
    R[0] = c; R[1] = -s; R[2] = 0; R[3] = s; R[4] = c; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    if (theta < 0.00001) {
      for (int i = 0; i < 9; i++) V[i] = R[i];
    } else {
      double st, ct;
      se3_sincos(theta, st, ct);
      const double a = (1 - ct) / (theta * theta), b = (theta - st) / se3_cube(theta);
      for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
    }
  } else if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    double st, ct;
    se3_sincos(theta, st, ct);
    const double a = st / theta, b = (1 - ct) / (theta * theta), c = (theta - st) / se3_cube(theta);
    for (int i = 0; i < 9; i++) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  double t[3];
  for (int r = 0; r < 3; r++) t[r] = V[r * 3] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
  return se3_from_Rt(R, t);
}
// Converter::toSE3Quat(cv::Mat float 4x4) (src/Converter.cc:37-47)
PS_HD Se3 se3_from_mat4f(const float* m) {
  double R[9], t[3];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = (double)m[r * 4 + c]; t[r] = (double)m[r * 4 + 3]; }
  return se3_from_Rt(R, t);
}
// Converter::toCvMat(SE3Quat) via to_homogeneous_matrix (src/Converter.cc:49-71, se3quat.h:296-304)
PS_HD void se3_to_mat4f(const Se3& T, float* m) {
  double R[9];
  se3_quat_to_R(T.q, R);
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) m[r * 4 + c] = (float)R[r * 3 + c]; m[r * 4 + 3] = (float)T.t[r]; }
  m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}
