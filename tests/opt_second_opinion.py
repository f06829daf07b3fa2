"""An independent statement of the two pose-only optimisers in numpy - test infrastructure, written from the reference's text and the vendored
g2o it drives, with its OWN residuals, Jacobians, SE3 algebra and dense 6 x 6 solves; no code shared with oracle/ (the
existing dense-LM second opinion covered the object BA only and took its edge arithmetic from the restatement).

  pose_optimization   Optimizer::PoseOptimization            /root/reference/src/Optimizer.cc:249-477
  cfse3_optimization  Optimizer::CFSE3ObjStateOptimization   :479-753 + EdgeTransConstraintFromDetction include/g2o_Object.h:407-422
  edges               Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:266-360 (error, analytic Jacobians; the stereo edge's `float invz`)
  Levenberg-Marquardt Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189 (lambda init tau = 1e-5, gain ratio, 1/3 .. 2/3
                      scaling, at most 10 trials, Raul's stop rule), sparse_optimizer.cpp:100-114 (activeRobustChi2), :354-419 (optimize)
  quadratic form      core/base_unary_edge.hpp:43-73, robust_kernel_impl.cpp:78-91 (Huber), base_unary_edge.hpp:83-121 (numeric Jacobian)
  SE3                 types/se3quat.h (exp with its theta < 1e-5 branch, operator*, normalizeRotation, map)

What is reproduced because it decides results: the estimate is reset every round in PoseOptimization and never in CFSE3; the classification
between rounds reads every ACTIVE edge's error where the last computeActiveErrors left it - possibly a rejected trial's estimate - and
recomputes only the edges that sat the round out; the comparison is `float chi2 > 5.991f / 7.815f`; the robust kernels of the projection
edges go after the third round, the prior's never; the increment vector survives a failed factorisation."""
import numpy as np

DELTA_MONO = float(np.float32(np.sqrt(5.991)))       # const float deltaMono = sqrt(5.991)
DELTA_STEREO = float(np.float32(np.sqrt(7.815)))
CHI2_MONO, CHI2_STEREO = np.float32(5.991), np.float32(7.815)


# ---- SE3 as (unit quaternion xyzw, translation) -------------------------------------------------------------------------
def quat_from_matrix(R):
    """Eigen::Quaterniond(Matrix3d) (Shepperd's branches as Eigen takes them)"""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t; q[1] = (R[0, 2] - R[2, 0]) * t; q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def quat_to_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_rotate(q, v):
    """q * v for one vector or rows of vectors (the rotation matrix of the unit quaternion applied to v)"""
    return np.asarray(v) @ quat_to_matrix(q).T


def normalize_rotation(q):
    if q[3] < 0:
        q = -q
    return q / np.sqrt(q @ q)


def se3_mul(a, b):
    """SE3Quat::operator*: t = ta + ra tb, r = ra rb, normalizeRotation"""
    return normalize_rotation(quat_mul(a[0], b[0])), a[1] + quat_rotate(a[0], b[1])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], float)


def se3_exp(u):
    """SE3Quat::exp: update = (omega, upsilon)"""
    omega, ups = np.asarray(u[:3], float), np.asarray(u[3:], float)
    theta = np.sqrt(omega @ omega)
    Om = skew(omega)
    if theta < 0.00001:
        R = np.eye(3) + Om + Om @ Om
        V = R
    else:
        Om2 = Om @ Om
        R = np.eye(3) + np.sin(theta) / theta * Om + (1 - np.cos(theta)) / (theta * theta) * Om2
        V = np.eye(3) + (1 - np.cos(theta)) / (theta * theta) * Om + (theta - np.sin(theta)) / theta ** 3 * Om2
    return quat_from_matrix(R), V @ ups


def se3_from_mat4f(T):
    """Converter::toSE3Quat: the float matrix widened to double, Quaterniond(R), t"""
    T = np.asarray(T, np.float32).astype(np.float64)
    return quat_from_matrix(T[:3, :3]), T[:3, 3].copy()


def se3_to_mat4f(p):
    """Converter::toCvMat(SE3Quat): to_homogeneous_matrix narrowed to float"""
    T = np.eye(4)
    T[:3, :3] = quat_to_matrix(p[0]); T[:3, 3] = p[1]
    return T.astype(np.float32)


# ---- the two projection edges, vectorised over the edges of one vertex --------------------------------------------------------
def edge_errors(pose, X, obs, mono, K):
    fx, fy, cx, cy, bf = K
    p = quat_rotate(pose[0], X) + pose[1]
    e = np.zeros((len(X), 3))
    m = mono
    # EdgeSE3ProjectXYZOnlyPose: project2d = x / z, y / z in double
    e[m, 0] = obs[m, 0] - (p[m, 0] / p[m, 2] * fx + cx)
    e[m, 1] = obs[m, 1] - (p[m, 1] / p[m, 2] * fy + cy)
    # EdgeStereoSE3ProjectXYZOnlyPose::cam_project: const float invz = 1.0f / trans_xyz[2]
    s = ~m
    invz = (1.0 / p[s, 2]).astype(np.float32).astype(np.float64)
    u = p[s, 0] * invz * fx + cx
    e[s, 0] = obs[s, 0] - u
    e[s, 1] = obs[s, 1] - (p[s, 1] * invz * fy + cy)
    e[s, 2] = obs[s, 2] - (u - bf * invz)
    return p, e


def edge_jacobians(p, mono, K):
    fx, fy, _, _, bf = K
    x, y = p[:, 0], p[:, 1]
    invz = 1.0 / p[:, 2]
    invz2 = invz * invz
    J = np.zeros((len(p), 3, 6))
    J[:, 0, 0] = x * y * invz2 * fx; J[:, 0, 1] = -(1 + x * x * invz2) * fx; J[:, 0, 2] = y * invz * fx
    J[:, 0, 3] = -invz * fx; J[:, 0, 5] = x * invz2 * fx
    J[:, 1, 0] = (1 + y * y * invz2) * fy; J[:, 1, 1] = -x * y * invz2 * fy; J[:, 1, 2] = -x * invz * fy
    J[:, 1, 4] = -invz * fy; J[:, 1, 5] = y * invz2 * fy
    J[:, 2, 0] = J[:, 0, 0] - bf * y * invz2; J[:, 2, 1] = J[:, 0, 1] + bf * x * invz2; J[:, 2, 2] = J[:, 0, 2]
    J[:, 2, 3] = J[:, 0, 3]; J[:, 2, 5] = J[:, 0, 5] - bf * invz2
    J[mono, 2, :] = 0
    return J


def huber(chi2, delta):
    """RobustKernelHuber::robustify: (rho, rho') per edge"""
    dsqr = delta * delta
    inl = chi2 <= dsqr
    sq = np.sqrt(np.where(inl, 1.0, chi2))
    return np.where(inl, chi2, 2 * sq * delta - dsqr), np.where(inl, 1.0, delta / sq)


class _Vertex:
    def __init__(self, pose, X, obs, is2, K, prior=None):
        self.pose = pose
        self.X, self.obs, self.is2, self.K = X, obs, is2, K
        self.mono = obs[:, 2] < 0
        self.delta = np.where(self.mono, DELTA_MONO, DELTA_STEREO)
        self.level1 = np.zeros(len(X), bool)
        self.chi2 = np.zeros(len(X))              # e->chi2() where the last computeError left it
        self.prior = prior                          # measurement of the translation prior (CFSE3) or None
        self.prior_err = None

    def errors(self, pose, robust):
        """computeActiveErrors on this vertex's active edges at `pose`; returns their robust chi2 sum"""
        act = ~self.level1
        _, e = edge_errors(pose, self.X[act], self.obs[act], self.mono[act], self.K)
        c = (e * e).sum(1) * self.is2[act]
        self.chi2[act] = c
        total = (huber(c, self.delta[act])[0] if robust else c).sum()
        if self.prior is not None:
            self.prior_err = self.prior - pose[1]
            total += huber(np.array([50.0 * (self.prior_err @ self.prior_err)]), DELTA_MONO)[0][0]
        return total

    def build(self, robust):
        """linearizeOplus + constructQuadraticForm of the active edges at the current estimate -> (H 6x6, b 6)"""
        act = ~self.level1
        p, e = edge_errors(self.pose, self.X[act], self.obs[act], self.mono[act], self.K)
        J = edge_jacobians(p, self.mono[act], self.K)
        w = self.is2[act]
        rho1 = huber((e * e).sum(1) * w, self.delta[act])[1] if robust else np.ones(len(w))
        wo = rho1 * w
        H = np.einsum("n,nri,nrj->ij", wo, J, J)
        b = -np.einsum("n,nri,nr->i", wo, J, e)
        if self.prior is not None:
            # numeric Jacobian through oplus, central differences with delta = 1e-9
            err = self.prior - self.pose[1]
            Jp = np.zeros((3, 6))
            for d in range(6):
                add = np.zeros(6)
                add[d] = 1e-9
                ep = self.prior - se3_mul(se3_exp(add), self.pose)[1]
                add[d] = -1e-9
                em = self.prior - se3_mul(se3_exp(add), self.pose)[1]
                Jp[:, d] = (1.0 / (2 * 1e-9)) * (ep - em)
            r1 = huber(np.array([50.0 * (err @ err)]), DELTA_MONO)[1][0]
            H = H + r1 * 50.0 * (Jp.T @ Jp)
            b = b - r1 * 50.0 * (Jp.T @ err)
        return H, b

    def classify(self, robust_unused=None):
        """Optimizer.cc:404-466: edges that sat out get computeError() at the final estimate, the others keep their cached error"""
        out = self.level1
        if out.any():
            _, e = edge_errors(self.pose, self.X[out], self.obs[out], self.mono[out], self.K)
            self.chi2[out] = (e * e).sum(1) * self.is2[out]
        c = self.chi2.astype(np.float32)
        self.level1 = c > np.where(self.mono, CHI2_MONO, CHI2_STEREO)
        return int(self.level1.sum())


def _lm(verts, robust, iterations=10, trace=None):
    """SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg over the vertices' block-diagonal system"""
    lam = ni = 0.0
    nbad = 0
    x = [np.zeros(6) for _ in verts]                       # the solver's increment vector persists over trials and iterations
    for it in range(iterations):
        current = sum(v.errors(v.pose, robust) for v in verts)
        ini = current
        Hb = [v.build(robust) for v in verts]
        if it == 0:
            lam = 1e-5 * max(np.abs(np.diag(H)).max() for H, _ in Hb)
            ni = 2.0
            nbad = 0
        rho = 0.0
        qmax = 0
        while True:
            ok = True
            sols = []
            for H, b in Hb:
                A = H + lam * np.eye(6)
                try:
                    np.linalg.cholesky(A)
                    sols.append(np.linalg.solve(A, b))
                except np.linalg.LinAlgError:
                    ok = False
            if ok:
                x = sols
            trial = [se3_mul(se3_exp(xi), v.pose) for xi, v in zip(x, verts)]
            temp = sum(v.errors(p, robust) for v, p in zip(verts, trial))
            if not ok:
                temp = np.finfo(float).max
            scale = sum(float(xi @ (lam * xi + b)) for xi, (_, b) in zip(x, Hb)) + 1e-3
            rho = (current - temp) / scale
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                current = temp
                for v, p in zip(verts, trial):
                    v.pose = p
            else:
                lam *= ni
                ni *= 2
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        if trace is not None:
            trace.append((current, lam, qmax))
        if qmax == 10 or rho == 0:
            break
        if (ini - current) * 1e3 < ini:
            nbad += 1
        else:
            nbad = 0
        if nbad >= 3:
            break


def pose_optimization(p):
    """p: a synth.pose_problem dict -> (return value, Tcw float32 4x4, outlier mask uint8[n], trace [(chi2, lambda, trials)])"""
    n = len(p["xw"])
    valid = np.asarray(p["valid"]).astype(bool)
    idx = np.nonzero(valid)[0]
    outlier = np.asarray(p.get("outlier0", np.zeros(n, np.uint8)), np.uint8).copy()
    tcw0 = np.asarray(p["tcw0"], np.float32)
    if len(idx) < 15:
        return 0, tcw0, outlier, []
    K = [float(v) for v in p["K"]]
    v = _Vertex(se3_from_mat4f(tcw0), np.asarray(p["xw"], np.float32)[idx].astype(np.float64), np.asarray(p["obs"], np.float32)[idx].astype(np.float64),
                np.asarray(p["inv_sigma2"], np.float32)[idx].astype(np.float64), K)
    trace = []
    robust = True
    nbad = 0
    for rnd in range(4):
        v.pose = se3_from_mat4f(tcw0)                      # vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw)) every round
        if (~v.level1).any():
            _lm([v], robust, 10, trace)
        nbad = v.classify()
        if rnd == 2:
            robust = False
    outlier[idx] = v.level1.astype(np.uint8)
    return len(idx) - nbad, se3_to_mat4f(v.pose), outlier, trace


def cfse3_optimization(objs, K):
    """objs: [{xo, obs, inv_sigma2, valid, pose7 (t, q)}] -> (ok, poses7 [k, 7], outlier masks)"""
    K = [float(v) for v in K]
    verts, idxs = [], []
    total = 0
    for o in objs:
        valid = np.asarray(o["valid"]).astype(bool)
        idx = np.nonzero(valid)[0]
        p7 = np.asarray(o["pose7"], float)
        pose = (p7[3:7].copy(), p7[:3].copy())
        verts.append(_Vertex(pose, np.asarray(o["xo"], np.float32)[idx].astype(np.float64), np.asarray(o["obs"], np.float32)[idx].astype(np.float64),
                             np.asarray(o["inv_sigma2"], np.float32)[idx].astype(np.float64), K, prior=p7[:3].copy()))
        idxs.append(idx)
        total += len(idx) + 1
    outs = [np.zeros(len(o["xo"]), np.uint8) for o in objs]
    if not objs or total < 15:
        return 0, np.array([o["pose7"] for o in objs], float).reshape(len(objs), 7), outs
    robust = True
    for rnd in range(4):
        _lm(verts, robust, 10)
        for v in verts:
            v.classify()
        if rnd == 2:
            robust = False
    for v, idx, out in zip(verts, idxs, outs):
        out[idx] = v.level1.astype(np.uint8)
    return 1, np.array([np.concatenate([v.pose[1], v.pose[0]]) for v in verts]), outs
