// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header; the product never links this).
//
// CPU restatement (FP64, no Eigen / g2o) of the reference's optimisation back-end:
//   Optimizer::PoseOptimization              /root/reference/src/Optimizer.cc:249-477
//   Optimizer::CFSE3ObjStateOptimization     :479-753
//   Optimizer::ObjectLocalBundleAdjustment   :755-1075 (graph already collected into arrays)
// and of the g2o machinery those three drive (Thirdparty/g2o/g2o/...):
//   types/se3quat.h:58-60,104-125,248-279,306-311      SE3Quat ctor / operator* / map / exp / normalizeRotation
//   types/types_six_dof_expmap.{h,cpp}                 the four projection edges (errors + analytic Jacobians)
//   src/g2o_Object.cc:26-56,190-213                    exptwist_norollpitch, VertexSE3Fix::oplusImpl
//   include/g2o_Object.h:407-422                       EdgeTransConstraintFromDetction (numeric Jacobian,
//                                                      core/base_unary_edge.hpp:83-121, delta 1e-9)
//   core/robust_kernel_impl.cpp:78-91                  Huber
//   core/base_unary_edge.hpp:43-73, base_binary_edge.hpp:55-120, base_edge.h:96-102   quadratic forms
//   core/optimization_algorithm_levenberg.cpp:61-189   LM control (lambda, gain ratio, 10 trials, stop rule)
//   core/block_solver.hpp:354-485,502-608              Schur complement, back-substitution, setLambda
//   core/sparse_optimizer.cpp:61-114,354-435           active errors / robust chi2 / update
//   solvers/linear_solver_dense.h, linear_solver_eigen.h  (restated as an unpivoted dense LDL^T)
//   src/Converter.cc:37-71                             float cv::Mat <-> SE3Quat
//
// PARITY UNPINNED by the reference (it has no tests; Eigen is not available here).  Pins used instead
// (tests/test_oracle_opt.py): exp/log round trips, analytic-vs-numeric Jacobians, Schur == full dense
// solve, recovery of the generating pose on noise-free data, and an independent numpy LM.
// Documented deviations: Eigen's pivoted LDLT / SimplicialLDLT are replaced by an unpivoted LDL^T
// (identical solutions for the positive-definite damped systems LM produces); FP64 sums run in edge
// order like g2o's (optimizable_graph.h:112-117), the GPU path may reduce in another order.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct SE3 { double q[4]; double t[3]; };  // q = (x, y, z, w)

void quat_from_R(const double R[9], double q[4]) {  // Eigen::Quaterniond(Matrix3d): Shepperd
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * s;
    s = 0.5 / s;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * s;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
  }
}
void normalize_rotation(SE3& T) {  // se3quat.h:306-311
  if (T.q[3] < 0) for (int i = 0; i < 4; i++) T.q[i] = -T.q[i];
  double n = std::sqrt(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
  for (int i = 0; i < 4; i++) T.q[i] /= n;
}
void quat_to_R(const double q[4], double R[9]) {  // Eigen toRotationMatrix
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
void quat_rotate(const double q[4], const double v[3], double o[3]) {  // Eigen _transformVector
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (int i = 0; i < 3; i++) uv[i] += uv[i];
  o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
SE3 se3_from_Rt(const double R[9], const double t[3]) {  // SE3Quat(R, t), se3quat.h:58-60
  SE3 T;
  quat_from_R(R, T.q);
  for (int i = 0; i < 3; i++) T.t[i] = t[i];
  normalize_rotation(T);
  return T;
}
SE3 se3_mul(const SE3& a, const SE3& b) {  // se3quat.h:104-110
  SE3 r;
  double rt[3];
  quat_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  quat_mul(a.q, b.q, r.q);
  normalize_rotation(r);
  return r;
}
void se3_map(const SE3& T, const double x[3], double o[3]) {  // se3quat.h:242-245
  quat_rotate(T.q, x, o);
  for (int i = 0; i < 3; i++) o[i] += T.t[i];
}
SE3 se3_from_mat4f(const float* m) {  // Converter::toSE3Quat, Converter.cc:37-47
  double R[9], t[3];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = m[r * 4 + c]; t[r] = m[r * 4 + 3]; }
  return se3_from_Rt(R, t);
}
void se3_to_mat4f(const SE3& T, float* m) {  // Converter::toCvMat(SE3Quat), Converter.cc:49-71
  double R[9];
  quat_to_R(T.q, R);
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) m[r * 4 + c] = (float)R[r * 3 + c]; m[r * 4 + 3] = (float)T.t[r]; }
  m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1;
}
void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
void skew(const double v[3], double M[9]) {
  M[0] = 0; M[1] = -v[2]; M[2] = v[1]; M[3] = v[2]; M[4] = 0; M[5] = -v[0]; M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
// SE3Quat::exp (se3quat.h:248-279); norollpitch = exptwist_norollpitch (g2o_Object.cc:26-56)
SE3 se3_exp(const double u[6], bool norollpitch) {
  double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], V[9];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  skew(omega, Om);
  mat3_mul(Om, Om, Om2);
  if (norollpitch) {
    const double c = std::cos(omega[2]), s = std::sin(omega[2]);
    const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    std::memcpy(R, Rz, sizeof(R));
    if (theta < 0.00001) std::memcpy(V, R, sizeof(V));
    else {
      const double a = (1 - std::cos(theta)) / (theta * theta), b = (theta - std::sin(theta)) / std::pow(theta, 3);
      for (int i = 0; i < 9; i++) V[i] = I[i] + a * Om[i] + b * Om2[i];
    }
  } else if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) R[i] = I[i] + Om[i] + Om2[i];
    std::memcpy(V, R, sizeof(V));
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 9; i++) { R[i] = I[i] + a * Om[i] + b * Om2[i]; V[i] = I[i] + b * Om[i] + c * Om2[i]; }
  }
  double t[3];
  for (int r = 0; r < 3; r++) t[r] = V[r * 3] * ups[0] + V[r * 3 + 1] * ups[1] + V[r * 3 + 2] * ups[2];
  return se3_from_Rt(R, t);
}
// SE3Quat::log (se3quat.h:203-240) — used by the tests only
void se3_log(const SE3& T, double out[6]) {
  double R[9];
  quat_to_R(T.q, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]}, omega[3], Om[9], Om2[9], Vinv[9];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d > 0.99999) {
    for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
    skew(omega, Om); mat3_mul(Om, Om, Om2);
    for (int i = 0; i < 9; i++) Vinv[i] = I[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    const double theta = std::acos(d);
    for (int i = 0; i < 3; i++) omega[i] = theta / (2 * std::sqrt(1 - d * d)) * dR[i];
    skew(omega, Om); mat3_mul(Om, Om, Om2);
    const double c = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vinv[i] = I[i] - 0.5 * Om[i] + c * Om2[i];
  }
  for (int i = 0; i < 3; i++) { out[i] = omega[i]; out[i + 3] = Vinv[i * 3] * T.t[0] + Vinv[i * 3 + 1] * T.t[1] + Vinv[i * 3 + 2] * T.t[2]; }
}

// ---------------------------------------------------------------------------------------------------------
enum { E_MONO_POSE = 0, E_STEREO_POSE = 1, E_TRANS_PRIOR = 2, E_MONO_BA = 3, E_STEREO_BA = 4 };

struct Edge {
  int type, pose, point;      // point = -1 for unary edges
  double X[3];                // fixed 3-D point of unary projection edges
  double obs[3];
  double info;                // information = info * I
  double delta;               // Huber delta (already rounded through float like the reference's const float)
  bool robust;
  int level;
  double err[3];
  int dim;
};

struct Problem {
  double fx, fy, cx, cy, bf;
  std::vector<SE3> poses;
  std::vector<uint8_t> pose_fixed, pose_norollpitch;
  std::vector<double> points;   // 3 per point
  std::vector<Edge> edges;
};

struct IterTrace { double chi2, lambda; int trials; };

double edge_chi2(const Edge& e) {
  double s = 0;
  for (int i = 0; i < e.dim; i++) s += e.err[i] * e.err[i];
  return s * e.info;   // information is a scaled identity: err^T (info I) err
}
void huber(double e, double delta, double rho[3]) {  // robust_kernel_impl.cpp:78-91
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
  else { const double sq = std::sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}

void compute_error(const Problem& P, Edge& e) {
  const SE3& T = P.poses[e.pose];
  if (e.type == E_TRANS_PRIOR) { for (int i = 0; i < 3; i++) e.err[i] = e.obs[i] - T.t[i]; return; }
  const double* X = e.point >= 0 ? &P.points[3 * e.point] : e.X;
  double p[3];
  se3_map(T, X, p);
  if (e.type == E_MONO_POSE || e.type == E_MONO_BA) {  // project2d then fx, cx
    e.err[0] = e.obs[0] - (p[0] / p[2] * P.fx + P.cx);
    e.err[1] = e.obs[1] - (p[1] / p[2] * P.fy + P.cy);
  } else {  // cam_project with float invz (types_six_dof_expmap.cpp:151,301)
    const float invz = (float)(1.0f / p[2]);
    const double u = p[0] * invz * P.fx + P.cx, v = p[1] * invz * P.fy + P.cy;
    e.err[0] = e.obs[0] - u; e.err[1] = e.obs[1] - v; e.err[2] = e.obs[2] - (u - P.bf * invz);
  }
}
bool depth_positive(const Problem& P, const Edge& e) {
  double p[3];
  se3_map(P.poses[e.pose], e.point >= 0 ? &P.points[3 * e.point] : e.X, p);
  return p[2] > 0.0;
}
SE3 oplus_pose(const Problem& P, int i, const double u[6]) {
  if (P.pose_norollpitch[i]) {  // VertexSE3Fix::oplusImpl with whether_fixrollpitch (g2o_Object.cc:190-213)
    double u2[6] = {0, 0, u[2], u[3], u[4], u[5]};
    return se3_mul(se3_exp(u2, true), P.poses[i]);
  }
  return se3_mul(se3_exp(u, false), P.poses[i]);
}
// Jacobians: Jp (dim x 6, pose, cols 0-2 rotation 3-5 translation), Jx (dim x 3, point)
void linearize(Problem& P, Edge& e, double Jp[18], double Jx[9]) {
  std::memset(Jp, 0, sizeof(double) * 18);
  std::memset(Jx, 0, sizeof(double) * 9);
  if (e.type == E_TRANS_PRIOR) {  // numeric central differences, delta = 1e-9 (base_unary_edge.hpp:83-121)
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    double keep[3] = {e.err[0], e.err[1], e.err[2]};
    const SE3 backup = P.poses[e.pose];
    for (int d = 0; d < 6; d++) {
      double add[6] = {0, 0, 0, 0, 0, 0}, e1[3];
      add[d] = delta;
      P.poses[e.pose] = oplus_pose(P, e.pose, add);
      compute_error(P, e);
      std::memcpy(e1, e.err, sizeof(e1));
      P.poses[e.pose] = backup;
      add[d] = -delta;
      P.poses[e.pose] = oplus_pose(P, e.pose, add);
      compute_error(P, e);
      P.poses[e.pose] = backup;
      for (int r = 0; r < 3; r++) Jp[r * 6 + d] = scalar * (e1[r] - e.err[r]);
    }
    std::memcpy(e.err, keep, sizeof(keep));
    return;
  }
  const SE3& T = P.poses[e.pose];
  const double* X = e.point >= 0 ? &P.points[3 * e.point] : e.X;
  double p[3];
  se3_map(T, X, p);
  const double x = p[0], y = p[1], fx = P.fx, fy = P.fy, bf = P.bf;
  if (e.type == E_MONO_POSE || e.type == E_STEREO_POSE) {  // types_six_dof_expmap.cpp:266-360
    const double invz = 1.0 / p[2], invz_2 = invz * invz;
    Jp[0] = x * y * invz_2 * fx; Jp[1] = -(1 + (x * x * invz_2)) * fx; Jp[2] = y * invz * fx;
    Jp[3] = -invz * fx; Jp[4] = 0; Jp[5] = x * invz_2 * fx;
    Jp[6] = (1 + y * y * invz_2) * fy; Jp[7] = -x * y * invz_2 * fy; Jp[8] = -x * invz * fy;
    Jp[9] = 0; Jp[10] = -invz * fy; Jp[11] = y * invz_2 * fy;
    if (e.type == E_STEREO_POSE) {
      Jp[12] = Jp[0] - bf * y * invz_2; Jp[13] = Jp[1] + bf * x * invz_2; Jp[14] = Jp[2];
      Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - bf * invz_2;
    }
    return;
  }
  double R[9];
  quat_to_R(T.q, R);
  const double z = p[2], z_2 = z * z;
  if (e.type == E_MONO_BA) {  // EdgeSE3ProjectXYZ::linearizeOplus, types_six_dof_expmap.cpp:103-139
    const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 3; c++)
        Jx[r * 3 + c] = -1. / z * (tmp[r * 3] * R[c] + tmp[r * 3 + 1] * R[3 + c] + tmp[r * 3 + 2] * R[6 + c]);
  } else {  // EdgeStereoSE3ProjectXYZ::linearizeOplus, :188-232
    for (int c = 0; c < 3; c++) {
      Jx[c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
      Jx[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
      Jx[6 + c] = Jx[c] - bf * R[6 + c] / z_2;
    }
  }
  Jp[0] = x * y / z_2 * fx; Jp[1] = -(1 + (x * x / z_2)) * fx; Jp[2] = y / z * fx;
  Jp[3] = -1. / z * fx; Jp[4] = 0; Jp[5] = x / z_2 * fx;
  Jp[6] = (1 + y * y / z_2) * fy; Jp[7] = -x * y / z_2 * fy; Jp[8] = -x / z * fy;
  Jp[9] = 0; Jp[10] = -1. / z * fy; Jp[11] = y / z_2 * fy;
  if (e.type == E_STEREO_BA) {
    Jp[12] = Jp[0] - bf * y / z_2; Jp[13] = Jp[1] + bf * x / z_2; Jp[14] = Jp[2];
    Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - bf / z_2;
  }
}

// unpivoted dense LDL^T solve of A x = b (n x n, row-major, symmetric); false when a pivot is not > 0
bool ldlt_solve(std::vector<double> A, int n, const double* b, double* x, bool need_positive) {
  std::vector<double> D(n);
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k] * D[k];
    if (need_positive ? !(d > 0) : d == 0) return false;
    D[j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k] * D[k];
      A[i * n + j] = s / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * n + k] * y[k]; y[i] = s; }
  for (int i = 0; i < n; i++) y[i] /= D[i];
  for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= A[k * n + i] * x[k]; x[i] = s; }
  return true;
}
bool inv3(const double M[9], double O[9]) {  // Eigen 3x3 inverse: adjugate / determinant
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  const double id = 1.0 / det;
  O[0] = c00 * id; O[1] = (M[2] * M[7] - M[1] * M[8]) * id; O[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  O[3] = c01 * id; O[4] = (M[0] * M[8] - M[2] * M[6]) * id; O[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  O[6] = c02 * id; O[7] = (M[1] * M[6] - M[0] * M[7]) * id; O[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  return det != 0;
}

// SparseOptimizer::initializeOptimization(level) + optimize(iterations) with OptimizationAlgorithmLevenberg
// on a BlockSolver_6_3 (Schur iff any active vertex is a point).  Returns the iterations performed.
int lm_optimize(Problem& P, int iterations, std::vector<IterTrace>* trace) {
  std::vector<int> act;
  for (size_t i = 0; i < P.edges.size(); i++) if (P.edges[i].level == 0) act.push_back((int)i);
  const int NP = (int)P.poses.size(), NL = (int)P.points.size() / 3;
  std::vector<int> pidx(NP, -1), lidx(NL, -1);
  for (int ei : act) {
    const Edge& e = P.edges[ei];
    if (!P.pose_fixed[e.pose]) pidx[e.pose] = 0;
    if (e.point >= 0) lidx[e.point] = 0;
  }
  int np = 0, nl = 0;
  for (int i = 0; i < NP; i++) if (pidx[i] == 0) pidx[i] = np++;
  for (int j = 0; j < NL; j++) if (lidx[j] == 0) lidx[j] = nl++;
  if (np + nl == 0) return -1;   // "0 vertices to optimize"
  const int sp = 6 * np, sl = 3 * nl, n = sp + sl;
  std::vector<double> Hpp((size_t)np * 36), Hll((size_t)nl * 9), W, b(n), x(n, 0.0);
  std::vector<uint8_t> Wset;
  const bool schur = nl > 0;
  if (schur) { W.assign((size_t)np * nl * 18, 0.0); Wset.assign((size_t)np * nl, 0); }
  double lambda = 0, ni = 2;
  int nBad = 0, done = 0;
  auto active_errors = [&]() { for (int ei : act) compute_error(P, P.edges[ei]); };
  auto robust_chi2 = [&]() {
    double chi = 0;
    for (int ei : act) {
      const Edge& e = P.edges[ei];
      if (e.robust) { double rho[3]; huber(edge_chi2(e), e.delta, rho); chi += rho[0]; }
      else chi += edge_chi2(e);
    }
    return chi;
  };
  for (int it = 0; it < iterations; it++) {
    active_errors();
    double currentChi = robust_chi2(), tempChi = currentChi;
    const double iniChi = currentChi;
    // ---- buildSystem ----
    std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
    if (schur) { std::fill(W.begin(), W.end(), 0.0); std::fill(Wset.begin(), Wset.end(), 0); }
    for (int ei : act) {
      Edge& e = P.edges[ei];
      double Jp[18], Jx[9];
      linearize(P, e, Jp, Jx);
      double w = e.info, rw = 1.0;
      if (e.robust) { double rho[3]; huber(edge_chi2(e), e.delta, rho); rw = rho[1]; }
      const double wo = rw * w;   // robustInformation = rho' * Omega
      const int pi = pidx[e.pose], li = e.point >= 0 ? lidx[e.point] : -1;
      if (pi >= 0) {
        for (int r = 0; r < 6; r++) {
          double s = 0;
          for (int d = 0; d < e.dim; d++) s += Jp[d * 6 + r] * w * e.err[d];
          b[6 * pi + r] -= rw * s;
          for (int c = 0; c < 6; c++) {
            double h = 0;
            for (int d = 0; d < e.dim; d++) h += Jp[d * 6 + r] * wo * Jp[d * 6 + c];
            Hpp[(size_t)pi * 36 + r * 6 + c] += h;
          }
        }
      }
      if (li >= 0) {
        for (int r = 0; r < 3; r++) {
          double s = 0;
          for (int d = 0; d < e.dim; d++) s += Jx[d * 3 + r] * w * e.err[d];
          b[sp + 3 * li + r] -= rw * s;
          for (int c = 0; c < 3; c++) {
            double h = 0;
            for (int d = 0; d < e.dim; d++) h += Jx[d * 3 + r] * wo * Jx[d * 3 + c];
            Hll[(size_t)li * 9 + r * 3 + c] += h;
          }
        }
        if (pi >= 0) {
          double* Wb = &W[((size_t)pi * nl + li) * 18];
          Wset[(size_t)pi * nl + li] = 1;
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 3; c++) {
              double h = 0;
              for (int d = 0; d < e.dim; d++) h += Jp[d * 6 + r] * wo * Jx[d * 3 + c];
              Wb[r * 3 + c] += h;
            }
        }
      }
    }
    if (it == 0) {  // computeLambdaInit: tau * max |diag| over every active vertex block
      double maxDiag = 0;
      for (int i = 0; i < np; i++) for (int j = 0; j < 6; j++) maxDiag = std::max(std::fabs(Hpp[(size_t)i * 36 + j * 7]), maxDiag);
      for (int i = 0; i < nl; i++) for (int j = 0; j < 3; j++) maxDiag = std::max(std::fabs(Hll[(size_t)i * 9 + j * 4]), maxDiag);
      lambda = 1e-5 * maxDiag;
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const std::vector<SE3> poses_backup = P.poses;
      const std::vector<double> points_backup = P.points;
      // ---- solve (H + lambda I) x = b ----
      bool ok2;
      if (!schur) {
        ok2 = true;   // block-diagonal system: LinearSolverDense factorises the whole matrix, so x is
        std::vector<double> xs(sp);   // only overwritten when every block is positive
        for (int i = 0; i < np && ok2; i++) {
          std::vector<double> A(Hpp.begin() + (size_t)i * 36, Hpp.begin() + (size_t)(i + 1) * 36);
          for (int j = 0; j < 6; j++) A[j * 7] += lambda;
          ok2 = ldlt_solve(A, 6, &b[6 * i], &xs[6 * i], true);
        }
        if (ok2) for (int j = 0; j < sp; j++) x[j] = xs[j];
      } else {
        std::vector<double> S((size_t)sp * sp, 0.0), bs(b.begin(), b.begin() + sp), Dinv((size_t)nl * 9);
        for (int i = 0; i < np; i++)
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) S[(size_t)(6 * i + r) * sp + 6 * i + c] = Hpp[(size_t)i * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
        for (int l = 0; l < nl; l++) {
          double D[9];
          for (int k = 0; k < 9; k++) D[k] = Hll[(size_t)l * 9 + k] + ((k % 4 == 0) ? lambda : 0.0);
          double* Di = &Dinv[(size_t)l * 9];
          inv3(D, Di);
          double db[3];
          for (int r = 0; r < 3; r++) db[r] = Di[r * 3] * b[sp + 3 * l] + Di[r * 3 + 1] * b[sp + 3 * l + 1] + Di[r * 3 + 2] * b[sp + 3 * l + 2];
          for (int i1 = 0; i1 < np; i1++) {
            if (!Wset[(size_t)i1 * nl + l]) continue;
            const double* B1 = &W[((size_t)i1 * nl + l) * 18];
            double BD[18];
            for (int r = 0; r < 6; r++)
              for (int c = 0; c < 3; c++) BD[r * 3 + c] = B1[r * 3] * Di[c] + B1[r * 3 + 1] * Di[3 + c] + B1[r * 3 + 2] * Di[6 + c];
            for (int r = 0; r < 6; r++) bs[6 * i1 + r] -= B1[r * 3] * db[0] + B1[r * 3 + 1] * db[1] + B1[r * 3 + 2] * db[2];
            for (int i2 = 0; i2 < np; i2++) {
              if (!Wset[(size_t)i2 * nl + l]) continue;
              const double* B2 = &W[((size_t)i2 * nl + l) * 18];
              for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++)
                  S[(size_t)(6 * i1 + r) * sp + 6 * i2 + c] -= BD[r * 3] * B2[c * 3] + BD[r * 3 + 1] * B2[c * 3 + 1] + BD[r * 3 + 2] * B2[c * 3 + 2];
            }
          }
        }
        std::vector<double> xp(sp > 0 ? sp : 1);
        ok2 = sp == 0 ? true : ldlt_solve(S, sp, bs.data(), xp.data(), false);
        if (ok2) {
          for (int i = 0; i < sp; i++) x[i] = xp[i];
          for (int l = 0; l < nl; l++) {  // xl = Dinv (bl - W^T xp)
            double c[3] = {b[sp + 3 * l], b[sp + 3 * l + 1], b[sp + 3 * l + 2]};
            for (int i1 = 0; i1 < np; i1++) {
              if (!Wset[(size_t)i1 * nl + l]) continue;
              const double* B1 = &W[((size_t)i1 * nl + l) * 18];
              for (int r = 0; r < 6; r++) for (int k = 0; k < 3; k++) c[k] -= B1[r * 3 + k] * xp[6 * i1 + r];
            }
            const double* Di = &Dinv[(size_t)l * 9];
            for (int r = 0; r < 3; r++) x[sp + 3 * l + r] = Di[r * 3] * c[0] + Di[r * 3 + 1] * c[1] + Di[r * 3 + 2] * c[2];
          }
        }
      }
      // ---- update (with whatever x holds, as g2o does), errors, gain ratio ----
      for (int i = 0; i < NP; i++) if (pidx[i] >= 0) P.poses[i] = oplus_pose(P, i, &x[6 * pidx[i]]);
      for (int j = 0; j < NL; j++) if (lidx[j] >= 0) for (int k = 0; k < 3; k++) P.points[3 * j + k] += x[sp + 3 * lidx[j] + k];
      active_errors();
      tempChi = robust_chi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < n; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scaleFactor = std::max(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        P.poses = poses_backup;
        P.points = points_backup;
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    done++;
    if (trace) trace->push_back(IterTrace{currentChi, lambda, qmax});
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  return done;
}

double huber_delta(double chi2_th) { return (double)(float)std::sqrt(chi2_th); }  // const float deltaX = sqrt(...)

// classification loop shared by PoseOptimization (Optimizer.cc:404-466) and CFSE3 (:652-725)
int classify(Problem& P, const std::vector<int>& edge_ids, uint8_t* outlier, const int* slot, int it) {
  int nBad = 0;
  for (int pass = 0; pass < 2; pass++)        // mono edges first, then stereo, like the reference
    for (int ei : edge_ids) {
      Edge& e = P.edges[ei];
      const bool mono = e.type == E_MONO_POSE;
      if ((pass == 0) != mono) continue;
      const int idx = slot[ei];
      if (outlier[idx]) compute_error(P, e);
      const float chi2 = (float)edge_chi2(e);
      const float th = mono ? 5.991f : 7.815f;
      if (chi2 > th) { outlier[idx] = 1; e.level = 1; nBad++; }
      else { outlier[idx] = 0; e.level = 0; }
      if (it == 2) e.robust = false;
    }
  return nBad;
}

}  // namespace

extern "C" {

// Optimizer::PoseOptimization.  obs = (u, v, uR) per keypoint, uR < 0 => monocular.  Returns the inlier count
// (0 when fewer than 15 correspondences).  trace (optional): up to 40 x (chi2, lambda, trials).
int orc_pose_optimize(int n, const float* xw, const float* obs, const float* inv_sigma2, const uint8_t* valid,
                      float fx, float fy, float cx, float cy, float bf, float* tcw16, uint8_t* outlier,
                      double* trace, int* ntrace) {
  Problem P;
  P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
  const SE3 T0 = se3_from_mat4f(tcw16);
  P.poses.push_back(T0);
  P.pose_fixed.push_back(0);
  P.pose_norollpitch.push_back(0);
  std::vector<int> slot, ids;
  int nInitial = 0;
  for (int i = 0; i < n; i++) {
    if (!valid[i]) continue;
    nInitial++;
    outlier[i] = 0;
    Edge e{};
    const bool mono = obs[3 * i + 2] < 0;
    e.type = mono ? E_MONO_POSE : E_STEREO_POSE;
    e.dim = mono ? 2 : 3;
    e.pose = 0; e.point = -1;
    for (int k = 0; k < 3; k++) { e.X[k] = xw[3 * i + k]; e.obs[k] = obs[3 * i + k]; }
    e.info = inv_sigma2[i];
    e.delta = huber_delta(mono ? 5.991 : 7.815);
    e.robust = true; e.level = 0;
    ids.push_back((int)P.edges.size());
    slot.push_back(i);
    P.edges.push_back(e);
  }
  if (ntrace) *ntrace = 0;
  if (nInitial < 15) return 0;
  int nBad = 0;
  std::vector<IterTrace> tr;
  for (int it = 0; it < 4; it++) {
    P.poses[0] = T0;
    lm_optimize(P, 10, &tr);
    nBad = classify(P, ids, outlier, slot.data(), it);
  }
  se3_to_mat4f(P.poses[0], tcw16);
  if (trace && ntrace) {
    *ntrace = (int)tr.size();
    for (size_t i = 0; i < tr.size(); i++) { trace[3 * i] = tr[i].chi2; trace[3 * i + 1] = tr[i].lambda; trace[3 * i + 2] = tr[i].trials; }
  }
  return nInitial - nBad;
}

// Optimizer::CFSE3ObjStateOptimization: k objects, object o owns points [off[o], off[o+1]).
// poses: k x 7 doubles (tx,ty,tz,qx,qy,qz,qw) in/out.  Returns 1 (true) or 0 (false: < 15 edges / no object).
int orc_cfse3_optimize(int k, const int* off, const float* xo, const float* obs, const float* inv_sigma2,
                       const uint8_t* valid, float fx, float fy, float cx, float cy, float bf, double* poses7,
                       uint8_t* outlier) {
  if (k == 0) return 0;
  Problem P;
  P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
  std::vector<int> slot, ids;
  for (int o = 0; o < k; o++) {
    SE3 T;
    for (int i = 0; i < 3; i++) T.t[i] = poses7[7 * o + i];
    for (int i = 0; i < 4; i++) T.q[i] = poses7[7 * o + 3 + i];
    P.poses.push_back(T);
    P.pose_fixed.push_back(0);
    P.pose_norollpitch.push_back(0);
  }
  int nTotal = 0;
  for (int o = 0; o < k; o++) {
    Edge pr{};
    pr.type = E_TRANS_PRIOR; pr.dim = 3; pr.pose = o; pr.point = -1;
    for (int i = 0; i < 3; i++) pr.obs[i] = P.poses[o].t[i];
    pr.info = 50; pr.delta = huber_delta(5.991); pr.robust = true; pr.level = 0;
    slot.push_back(-1);
    P.edges.push_back(pr);
    nTotal++;
    for (int i = off[o]; i < off[o + 1]; i++) {
      if (!valid[i]) continue;
      outlier[i] = 0;
      Edge e{};
      const bool mono = obs[3 * i + 2] < 0;
      e.type = mono ? E_MONO_POSE : E_STEREO_POSE;
      e.dim = mono ? 2 : 3;
      e.pose = o; e.point = -1;
      for (int c = 0; c < 3; c++) { e.X[c] = xo[3 * i + c]; e.obs[c] = obs[3 * i + c]; }
      e.info = inv_sigma2[i];
      e.delta = huber_delta(mono ? 5.991 : 7.815);
      e.robust = true; e.level = 0;
      ids.push_back((int)P.edges.size());
      slot.push_back(i);
      P.edges.push_back(e);
      nTotal++;
    }
  }
  if (nTotal < 15) return 0;
  for (int it = 0; it < 4; it++) {
    lm_optimize(P, 10, nullptr);
    // per object mono-then-stereo; the classification of an edge does not depend on the others
    classify(P, ids, outlier, slot.data(), it);
  }
  for (int o = 0; o < k; o++) {
    for (int i = 0; i < 3; i++) poses7[7 * o + i] = P.poses[o].t[i];
    for (int i = 0; i < 4; i++) poses7[7 * o + 3 + i] = P.poses[o].q[i];
  }
  return 1;
}

// Optimizer::ObjectLocalBundleAdjustment on a collected graph.
//   poses7 [np][7] in/out, pose_flags [np]: bit0 = fixed, bit1 = VertexSE3Fix{fix roll/pitch}
//   points [nl][3] in/out (object frame), edges: pose index, point index, obs (u,v,uR; uR<0 mono), invSigma2
//   erase [ne] out: 1 where the reference pushes the (KF, point) pair into vToErase
// Returns the number of erased observations.  trace: (chi2, lambda, trials) per LM iteration.
int orc_object_ba(int np, double* poses7, const uint8_t* pose_flags, int nl, double* points, int ne,
                  const int* e_pose, const int* e_point, const float* e_obs, const float* e_inv_sigma2,
                  float fx, float fy, float cx, float cy, float bf, uint8_t* erase, double* trace, int* ntrace) {
  Problem P;
  P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
  for (int i = 0; i < np; i++) {
    SE3 T;
    for (int c = 0; c < 3; c++) T.t[c] = poses7[7 * i + c];
    for (int c = 0; c < 4; c++) T.q[c] = poses7[7 * i + 3 + c];
    P.poses.push_back(T);
    P.pose_fixed.push_back(pose_flags[i] & 1);
    P.pose_norollpitch.push_back((pose_flags[i] >> 1) & 1);
  }
  P.points.assign(points, points + 3 * nl);
  for (int i = 0; i < ne; i++) {
    Edge e{};
    const bool mono = e_obs[3 * i + 2] < 0;
    e.type = mono ? E_MONO_BA : E_STEREO_BA;
    e.dim = mono ? 2 : 3;
    e.pose = e_pose[i]; e.point = e_point[i];
    for (int c = 0; c < 3; c++) e.obs[c] = e_obs[3 * i + c];
    e.info = e_inv_sigma2[i];
    e.delta = huber_delta(mono ? 5.991 : 7.815);
    e.robust = true; e.level = 0;
    P.edges.push_back(e);
  }
  std::vector<IterTrace> tr;
  lm_optimize(P, 5, &tr);
  for (Edge& e : P.edges) {   // Optimizer.cc:959-983
    const double th = e.type == E_MONO_BA ? 5.991 : 7.815;
    if (edge_chi2(e) > th || !depth_positive(P, e)) e.level = 1;
    e.robust = false;
  }
  lm_optimize(P, 10, &tr);
  int nerase = 0;
  for (int i = 0; i < ne; i++) {  // :988-1012
    const Edge& e = P.edges[i];
    const double th = e.type == E_MONO_BA ? 5.991 : 7.815;
    erase[i] = (edge_chi2(e) > th || !depth_positive(P, e)) ? 1 : 0;
    nerase += erase[i];
  }
  for (int i = 0; i < np; i++) {
    for (int c = 0; c < 3; c++) poses7[7 * i + c] = P.poses[i].t[c];
    for (int c = 0; c < 4; c++) poses7[7 * i + 3 + c] = P.poses[i].q[c];
  }
  std::memcpy(points, P.points.data(), sizeof(double) * 3 * nl);
  if (trace && ntrace) {
    *ntrace = (int)tr.size();
    for (size_t i = 0; i < tr.size(); i++) { trace[3 * i] = tr[i].chi2; trace[3 * i + 1] = tr[i].lambda; trace[3 * i + 2] = tr[i].trials; }
  }
  return nerase;
}

// helpers exposed for the tests
void orc_se3_exp(const double* u6, int norollpitch, double* out7) {
  SE3 T = se3_exp(u6, norollpitch != 0);
  for (int i = 0; i < 3; i++) out7[i] = T.t[i];
  for (int i = 0; i < 4; i++) out7[3 + i] = T.q[i];
}
void orc_se3_log(const double* in7, double* u6) {
  SE3 T;
  for (int i = 0; i < 3; i++) T.t[i] = in7[i];
  for (int i = 0; i < 4; i++) T.q[i] = in7[3 + i];
  se3_log(T, u6);
}
void orc_se3_from_mat4f(const float* m16, double* out7) {
  SE3 T = se3_from_mat4f(m16);
  for (int i = 0; i < 3; i++) out7[i] = T.t[i];
  for (int i = 0; i < 4; i++) out7[3 + i] = T.q[i];
}
void orc_se3_to_mat4f(const double* in7, float* m16) {
  SE3 T;
  for (int i = 0; i < 3; i++) T.t[i] = in7[i];
  for (int i = 0; i < 4; i++) T.q[i] = in7[3 + i];
  se3_to_mat4f(T, m16);
}
// error + analytic Jacobians of one projection edge (type 0/1 pose-only, 3/4 BA) at pose7 / point X
void orc_edge_eval(int type, const double* pose7, const double* X, const double* obs, double fx, double fy, double cx,
                   double cy, double bf, double* err3, double* Jp18, double* Jx9) {
  Problem P;
  P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
  SE3 T;
  for (int i = 0; i < 3; i++) T.t[i] = pose7[i];
  for (int i = 0; i < 4; i++) T.q[i] = pose7[3 + i];
  P.poses.push_back(T); P.pose_fixed.push_back(0); P.pose_norollpitch.push_back(0);
  Edge e{};
  e.type = type; e.dim = (type == E_MONO_POSE || type == E_MONO_BA) ? 2 : 3; e.pose = 0;
  if (type >= E_MONO_BA) { P.points.assign(X, X + 3); e.point = 0; } else { e.point = -1; for (int i = 0; i < 3; i++) e.X[i] = X[i]; }
  for (int i = 0; i < 3; i++) e.obs[i] = obs[i];
  compute_error(P, e);
  for (int i = 0; i < 3; i++) err3[i] = i < e.dim ? e.err[i] : 0;
  linearize(P, e, Jp18, Jx9);
}
void orc_huber(double e, double delta, double* rho3) { huber(e, delta, rho3); }


// The reprojection test of Tracking::DynamicStaticDiscrimination (/root/reference/src/Tracking.cc:2099-2181) for one detection:
// every object point is moved as if it were static (Pc = Tcw_cur * Tcw_last^-1 * Tco_last * Po), projected, and its chi-square
// against the current observation collected per kind (monocular / stereo); each list is sorted, values above 5 x median
// (element int(size / 2 + 0.5)) are dropped and the rest averaged.  Lists with fewer than 5 points keep the average 0.
// poses: (tx, ty, tz, qx, qy, qz, qw).  out4 = {monoDynaValAvg, stereoDynaValAvg}, outn = {monoPointNum, stereoPointNum}.
void orc_dynamic_discrimination(int n, const uint8_t* valid, const double* po, const float* obs, const float* inv_sigma2,
                                const double* last_tco7, const double* last_tcw7, const double* cur_tcw7, double fx, double fy,
                                double cx, double cy, float mbf, double* out_avg2, int* out_n2) {
  auto from7 = [](const double* v) { SE3 T; for (int i = 0; i < 3; i++) T.t[i] = v[i]; for (int i = 0; i < 4; i++) T.q[i] = v[3 + i]; return T; };
  const SE3 Tco = from7(last_tco7), Tl = from7(last_tcw7), Tc = from7(cur_tcw7);
  // SE3Quat::inverse (se3quat.h:112-117): r = conj, t = r * (t * -1)
  SE3 Tli;
  Tli.q[0] = -Tl.q[0]; Tli.q[1] = -Tl.q[1]; Tli.q[2] = -Tl.q[2]; Tli.q[3] = Tl.q[3];
  const double nt[3] = {Tl.t[0] * -1., Tl.t[1] * -1., Tl.t[2] * -1.};
  quat_rotate(Tli.q, nt, Tli.t);
  const SE3 Trel = se3_mul(Tc, Tli);                       // current_pose * last_pose.inverse()
  std::vector<double> monoDynaVal, stereoDynaVal;
  int monoPointNum = 0, stereoPointNum = 0;
  for (int j = 0; j < n; j++) {
    if (!valid[j]) continue;
    double Plc[3], Pc[3];
    se3_map(Tco, po + 3 * j, Plc);
    se3_map(Trel, Plc, Pc);
    const double invz = 1.0 / Pc[2];
    const double s = (double)inv_sigma2[j];
    const double z0 = cx + Pc[0] * invz * fx, z1 = cy + Pc[1] * invz * fy;
    const double e0 = (double)obs[3 * j] - z0, e1 = (double)obs[3 * j + 1] - z1;
    if (obs[3 * j + 2] < 0) {
      monoDynaVal.push_back(e0 * (s * e0) + e1 * (s * e1));
      monoPointNum++;
    } else {
      const double z2 = z0 - (double)mbf * invz;
      const double e2 = (double)obs[3 * j + 2] - z2;
      stereoDynaVal.push_back(e0 * (s * e0) + e1 * (s * e1) + e2 * (s * e2));
      stereoPointNum++;
    }
  }
  auto reject_and_average = [](std::vector<double>& v, int& num) {
    double avg = 0;
    if (num >= 5) {
      std::sort(v.begin(), v.end());
      const double median = v[int(v.size() / 2 + 0.5)];
      for (auto it = v.begin(); it != v.end();) {
        if ((*it) > 5 * median) { it = v.erase(it); num--; }   // (the reference relies on erase leaving `it` at the next element)
        else it++;
      }
      double sum = 0.0;
      for (double x : v) sum += x;                              // std::accumulate
      avg = sum / num;
    }
    return avg;
  };
  out_avg2[0] = reject_and_average(monoDynaVal, monoPointNum);
  out_avg2[1] = reject_and_average(stereoDynaVal, stereoPointNum);
  out_n2[0] = monoPointNum; out_n2[1] = stereoPointNum;
}

}  // extern "C"
