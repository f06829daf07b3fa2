"""An independent statement of Optimizer::ObjectLocalBundleAdjustment in numpy - test infrastructure, written from the reference's text and the
vendored g2o, with its OWN residuals, both Jacobian blocks, the constrained pose update and a dense solve of the FULL normal equations (no
Schur complement, no block structure: numpy.linalg.solve on (6 P + 3 L) unknowns); no code shared with oracle/.  VERDICT r05: the dense LM
that existed ran five iterations of one small graph on the restatement's own edge arithmetic.

  Optimizer::ObjectLocalBundleAdjustment   /root/reference/src/Optimizer.cc:755-1075 (graph, optimize(5), chi2 / depth pass that moves edges to
                                           level 1 and drops every robust kernel, optimize(10), erase list)
  EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ   Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:103-139, 188-232 (+ the stereo edge's float invz)
  VertexSE3Fix::oplusImpl, exptwist_norollpitch   src/g2o_Object.cc:190-213, 26-56 (omega_x = omega_y = 0, R = Rz(omega_z), V from the full Rodrigues form)
  VertexSBAPointXYZ::oplusImpl                    Thirdparty/g2o/g2o/types/types_sba.h:40-57 (estimate += update)
  quadratic form of a binary edge                 core/base_binary_edge.hpp:55-120 (fixed vertices get no block), Huber robust_kernel_impl.cpp:78-91
  Levenberg-Marquardt                             core/optimization_algorithm_levenberg.cpp:61-189, as tests/opt_second_opinion.py

The SE3 algebra is tests/opt_second_opinion.py's (the same author's independent code, not the restatement's)."""
import numpy as np

import opt_second_opinion as so

DELTA_MONO, DELTA_STEREO = so.DELTA_MONO, so.DELTA_STEREO


def exptwist_norollpitch(u):
    omega, ups = np.asarray(u[:3], float), np.asarray(u[3:], float)
    theta = np.sqrt(omega @ omega)
    c, s = np.cos(omega[2]), np.sin(omega[2])
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    if theta < 0.00001:
        V = R
    else:
        Om = so.skew(omega)
        V = np.eye(3) + (1 - np.cos(theta)) / (theta * theta) * Om + (theta - np.sin(theta)) / theta ** 3 * (Om @ Om)
    return so.quat_from_matrix(R), V @ ups


def oplus_pose(pose, u, norollpitch):
    if norollpitch:
        u = np.array(u, float)
        u[0] = u[1] = 0.0
        return so.se3_mul(exptwist_norollpitch(u), pose)
    return so.se3_mul(so.se3_exp(u), pose)


def edge(pose, X, obs, mono, K):
    """-> (error[3], J_pose[3, 6], J_point[3, 3], depth) of one projection edge (row 2 is zero for a monocular one)"""
    fx, fy, cx, cy, bf = K
    R = so.quat_to_matrix(pose[0])
    p = R @ X + pose[1]
    x, y, z = p
    z2 = z * z
    e = np.zeros(3)
    Jp = np.zeros((3, 6))
    Jx = np.zeros((3, 3))
    Jp[0] = [x * y / z2 * fx, -(1 + x * x / z2) * fx, y / z * fx, -1.0 / z * fx, 0.0, x / z2 * fx]
    Jp[1] = [(1 + y * y / z2) * fy, -x * y / z2 * fy, -x / z * fy, 0.0, -1.0 / z * fy, y / z2 * fy]
    if mono:
        e[0] = obs[0] - (x / z * fx + cx)
        e[1] = obs[1] - (y / z * fy + cy)
        tmp = np.array([[fx, 0.0, -x / z * fx], [0.0, fy, -y / z * fy]])
        Jx[:2] = -1.0 / z * tmp @ R
    else:
        invz = float(np.float32(1.0 / z))
        u = x * invz * fx + cx
        e[0] = obs[0] - u
        e[1] = obs[1] - (y * invz * fy + cy)
        e[2] = obs[2] - (u - bf * invz)
        Jx[0] = -fx * R[0] / z + fx * x * R[2] / z2
        Jx[1] = -fy * R[1] / z + fy * y * R[2] / z2
        Jx[2] = Jx[0] - bf * R[2] / z2
        Jp[2] = [Jp[0, 0] - bf * y / z2, Jp[0, 1] + bf * x / z2, Jp[0, 2], Jp[0, 3], 0.0, Jp[0, 5] - bf / z2]
    return e, Jp, Jx, z


def _huber(c, delta, robust):
    if not robust or c <= delta * delta:
        return c, 1.0
    sq = np.sqrt(c)
    return 2 * sq * delta - delta * delta, delta / sq


class Graph:
    def __init__(self, p):
        self.poses = [(np.asarray(q[3:7], float).copy(), np.asarray(q[:3], float).copy()) for q in np.asarray(p["poses"], float)]
        self.flags = np.asarray(p["pose_flags"]).astype(int)
        self.pts = np.asarray(p["points"], float).copy()
        self.ep, self.el = np.asarray(p["e_pose"]).astype(int), np.asarray(p["e_point"]).astype(int)
        self.obs = np.asarray(p["e_obs"], np.float32).astype(np.float64)
        self.is2 = np.asarray(p["e_inv_sigma2"], np.float32).astype(np.float64)
        self.mono = self.obs[:, 2] < 0
        self.K = [float(v) for v in p["K"]]
        self.level = np.zeros(len(self.ep), int)
        self.chi2 = np.zeros(len(self.ep))          # e->chi2() where the last computeActiveErrors left it

    def errors(self, poses, pts, robust):
        tot = 0.0
        for k in np.nonzero(self.level == 0)[0]:
            e, _, _, _ = edge(poses[self.ep[k]], pts[self.el[k]], self.obs[k], self.mono[k], self.K)
            c = self.is2[k] * (e @ e)
            self.chi2[k] = c
            tot += _huber(c, DELTA_MONO if self.mono[k] else DELTA_STEREO, robust)[0]
        return tot

    def optimize(self, iterations, robust, trace):
        act = np.nonzero(self.level == 0)[0]
        free = sorted({int(self.ep[k]) for k in act if not self.flags[self.ep[k]] & 1})
        lpts = sorted({int(self.el[k]) for k in act})
        if not act.size or not (free or lpts):
            return
        pi = {v: i for i, v in enumerate(free)}
        li = {v: i for i, v in enumerate(lpts)}
        sp = 6 * len(free)
        n = sp + 3 * len(lpts)
        lam = ni = 0.0
        nbad = 0
        x = np.zeros(n)
        for it in range(iterations):
            current = self.errors(self.poses, self.pts, robust)
            ini = current
            H, b = np.zeros((n, n)), np.zeros(n)
            for k in act:
                e, Jp, Jx, _ = edge(self.poses[self.ep[k]], self.pts[self.el[k]], self.obs[k], self.mono[k], self.K)
                w = self.is2[k] * _huber(self.is2[k] * (e @ e), DELTA_MONO if self.mono[k] else DELTA_STEREO, robust)[1]
                cols = []
                if self.ep[k] in pi:
                    cols.append((6 * pi[self.ep[k]], Jp))
                cols.append((sp + 3 * li[self.el[k]], Jx))
                for ca, Ja in cols:
                    b[ca:ca + Ja.shape[1]] -= w * (Ja.T @ e)
                    for cb, Jb in cols:
                        H[ca:ca + Ja.shape[1], cb:cb + Jb.shape[1]] += w * (Ja.T @ Jb)
            if it == 0:
                lam = 1e-5 * np.abs(np.diag(H)).max()
                ni = 2.0
                nbad = 0
            rho = 0.0
            q = 0
            while True:
                A = H + lam * np.eye(n)
                ok = True
                try:
                    np.linalg.cholesky(A)
                    x = np.linalg.solve(A, b)
                except np.linalg.LinAlgError:
                    ok = False
                tp = list(self.poses)
                for v, i in pi.items():
                    tp[v] = oplus_pose(self.poses[v], x[6 * i:6 * i + 6], bool(self.flags[v] & 2))
                tx = self.pts.copy()
                for v, i in li.items():
                    tx[v] = self.pts[v] + x[sp + 3 * i:sp + 3 * i + 3]
                temp = self.errors(tp, tx, robust)
                if not ok:
                    temp = np.finfo(float).max
                rho = (current - temp) / (float(x @ (lam * x + b)) + 1e-3)
                if rho > 0 and np.isfinite(temp):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni = 2.0
                    current = temp
                    self.poses, self.pts = tp, tx
                else:
                    lam *= ni
                    ni *= 2
                q += 1
                if not (rho < 0 and q < 10):
                    break
            trace.append((current, lam, q))
            if q == 10 or rho == 0:
                break
            nbad = nbad + 1 if (ini - current) * 1e3 < ini else 0
            if nbad >= 3:
                break

    def flagged(self):
        """chi2 above its threshold (double against the double literal) or the point behind the camera at the CURRENT estimates"""
        out = np.zeros(len(self.ep), bool)
        for k in range(len(self.ep)):
            z = edge(self.poses[self.ep[k]], self.pts[self.el[k]], self.obs[k], self.mono[k], self.K)[3]
            out[k] = self.chi2[k] > (5.991 if self.mono[k] else 7.815) or not (z > 0.0)
        return out


def object_local_bundle_adjustment(p):
    """-> (erased observations, poses7 [P, 7] (t, q), points [L, 3], erase mask, trace [(chi2, lambda, trials)])"""
    g = Graph(p)
    trace = []
    g.optimize(5, True, trace)
    g.level = g.flagged().astype(int)                    # ... and e->setRobustKernel(0) on every edge: the second run is not robust
    g.optimize(10, False, trace)
    erase = g.flagged()
    poses = np.array([np.concatenate([t, q]) for q, t in g.poses])
    return int(erase.sum()), poses, g.pts, erase.astype(np.uint8), trace
