"""An independent statement of the four order-dependent matchers, written from the reference's text in plain Python / numpy - test
infrastructure, no code shared with oracle/ (the pixel stages had a second author since r02, the greedy matchers - where a
single-author slip is most likely - had property tests only).

  search_by_bruce_matching      /root/reference/src/ORBmatcher.cc:2043-2155 (+ ComputeThreeMaxima :2658-2699, DescriptorDistance :2704-2720)
  search_by_projection_frame    :1613-1756   SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
  search_by_projection_points   :68-155      SearchByProjection(Frame& F, const vector<MapPoint*>&, th)
                                :157-248     SearchByProjection(Frame& F, nOrder, const vector<MapObjectPoint*>&, th)   (object=True)
  Frame::GetFeaturesInArea / GetObjectFeaturesInArea   /root/reference/src/Frame.cc:1808-1861, 1892-1951
  Frame::PosInGrid + AssignFeaturesToGrid              :2027-2037, 1636-1656 (a cell's list is in keypoint order)

Everything the reference holds as `float` is a numpy float32 here and every operation on it is rounded to float32 on its own; what the
reference leaves to the map data model comes in as flags, exactly as the C-ABI takes it (`occupied[j]`: the slot holds a point with
Observations() > 0; `observed[i]`: the candidate point has Observations() > 0, so its assignment blocks the slot for later candidates -
an unobserved one can be overwritten, "last assignment wins", and both count as matches).

One modelling question this file makes explicit: the camera-frame point of the frame-to-frame search is the cv::Mat expression
`Rcw*x3Dw+tcw` on CV_32F matrices.  `accumulate="double"` forms the three products and their sum in double and rounds once (cv::gemm's
general path, the restatement's choice); `accumulate="float"` rounds every product and every partial sum to float32 (the 3 x 3 small-matrix
path gemm takes for flags == 0 in OpenCV 2.4 / 3.x).  The results differ by at most an ulp of u, v; the tests run both and report whether a
match changes."""
import math

import numpy as np

F32 = np.float32
TH_HIGH, TH_LOW, TH_HIGH_FORDYNAMIC, HISTO_LENGTH = 100, 50, 130, 30
GRID_COLS, GRID_ROWS = 64, 48
_POP = np.array([bin(i).count("1") for i in range(256)], np.int64)


def descriptor_distance(a, b):
    """the SWAR popcount of the reference is a popcount: bits set in a ^ b over the 32 bytes"""
    return int(_POP[np.bitwise_xor(a, b)].sum())


def _distances(d, rows):
    """Hamming distance of descriptor d (32 bytes) to every row of `rows` [n, 32]"""
    return _POP[np.bitwise_xor(rows, d[None, :])].sum(axis=1)


def three_maxima(sizes):
    """ComputeThreeMaxima on the bins' sizes -> (ind1, ind2, ind3)"""
    max1 = max2 = max3 = 0
    ind1 = ind2 = ind3 = -1
    for i, s in enumerate(sizes):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            ind3, ind2, ind1 = ind2, ind1, i
        elif s > max2:
            max3, max2 = max2, s
            ind3, ind2 = ind2, i
        elif s > max3:
            max3 = s
            ind3 = i
    if F32(max2) < F32(0.1) * F32(max1):
        ind2 = ind3 = -1
    elif F32(max3) < F32(0.1) * F32(max1):
        ind3 = -1
    return ind1, ind2, ind3


def _c_round(v):
    """C round(): half away from zero"""
    return int(math.floor(abs(float(v)) + 0.5)) * (1 if v >= 0 else -1)


def _rot_bin(angle_from, angle_to):
    rot = F32(angle_from) - F32(angle_to)
    if rot < 0.0:
        rot = F32(rot + F32(360.0))
    b = _c_round(F32(rot * F32(F32(HISTO_LENGTH) / F32(360.0))))
    return 0 if b == HISTO_LENGTH else b


def search_by_bruce_matching(p, nnratio, check_ori):
    """-> (nmatches, query_of_train[nt]); a train's entry is the index of the last-frame point it was given to, -1 = NULL"""
    qd, td = np.asarray(p["q_desc"], np.uint8).reshape(-1, 32), np.asarray(p["t_desc"], np.uint8).reshape(-1, 32)
    qa, ta, qv = np.asarray(p["q_angle"], F32), np.asarray(p["t_angle"], F32), np.asarray(p["q_valid"])
    nt = len(td)
    out = [-1] * nt
    taken = np.zeros(nt, bool)
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    for i in range(len(qd)):
        if not qv[i]:
            continue
        best1, best2, bidx = 256, 256, -1
        if nt:
            d = _distances(qd[i], td)
            # the scan in train order, skipping the trains already given away: best = FIRST minimum, second = best of the rest
            for j in np.nonzero(~taken)[0]:
                dj = int(d[j])
                if dj < best1:
                    best2, best1, bidx = best1, dj, int(j)
                elif dj < best2:
                    best2 = dj
        if best1 <= TH_LOW and F32(best1) < F32(nnratio) * F32(best2):
            out[bidx] = i
            taken[bidx] = True
            if check_ori:
                hist[_rot_bin(qa[i], ta[bidx])].append(bidx)
            nm += 1
    if check_ori:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b in keep:
                continue
            for j in hist[b]:
                out[j] = -1
                nm -= 1
    return nm, np.array(out, np.int32) if nt else np.zeros(0, np.int32)


class Grid:
    """mGrid[ix][iy] of a frame: PosInGrid on every keypoint in order (a keypoint outside the grid is in no cell)"""

    def __init__(self, x, y, grid):
        self.x, self.y = np.asarray(x, F32), np.asarray(y, F32)
        self.min_x, self.min_y, self.gw_inv, self.gh_inv = [F32(v) for v in grid]
        self.cells = [[[] for _ in range(GRID_ROWS)] for _ in range(GRID_COLS)]
        for i in range(len(self.x)):
            px = _c_round(F32(F32(self.x[i] - self.min_x) * self.gw_inv))
            py = _c_round(F32(F32(self.y[i] - self.min_y) * self.gh_inv))
            if px < 0 or px >= GRID_COLS or py < 0 or py >= GRID_ROWS:
                continue
            self.cells[px][py].append(i)

    def features_in_area(self, x, y, r, octave, min_level=-1, max_level=-1):
        x, y, r = F32(x), F32(y), F32(r)
        out = []
        c0 = max(0, int(math.floor(F32(F32(F32(x - self.min_x) - r) * self.gw_inv))))
        if c0 >= GRID_COLS:
            return out
        c1 = min(GRID_COLS - 1, int(math.ceil(F32(F32(F32(x - self.min_x) + r) * self.gw_inv))))
        if c1 < 0:
            return out
        r0 = max(0, int(math.floor(F32(F32(F32(y - self.min_y) - r) * self.gh_inv))))
        if r0 >= GRID_ROWS:
            return out
        r1 = min(GRID_ROWS - 1, int(math.ceil(F32(F32(F32(y - self.min_y) + r) * self.gh_inv))))
        if r1 < 0:
            return out
        check = min_level > 0 or max_level >= 0
        for ix in range(c0, c1 + 1):
            for iy in range(r0, r1 + 1):
                for j in self.cells[ix][iy]:
                    if check:
                        if octave[j] < min_level:
                            continue
                        if max_level >= 0 and octave[j] > max_level:
                            continue
                    if abs(F32(self.x[j] - x)) < r and abs(F32(self.y[j] - y)) < r:
                        out.append(j)
        return out


def _mat3_vec_plus(T, v, accumulate):
    """rows 0..2 of the float 4 x 4 pose T: R v + t as the cv::Mat expression evaluates it (see the module text)"""
    out = []
    for r in range(3):
        if accumulate == "double":
            acc = float(T[r, 0]) * float(v[0]) + float(T[r, 1]) * float(v[1]) + float(T[r, 2]) * float(v[2])
            out.append(F32(F32(acc) + F32(T[r, 3])))
        else:
            acc = F32(F32(F32(T[r, 0]) * F32(v[0])) + F32(F32(T[r, 1]) * F32(v[1])))
            acc = F32(acc + F32(F32(T[r, 2]) * F32(v[2])))
            out.append(F32(acc + F32(T[r, 3])))
    return out


def search_by_projection_frame(pr, check_ori=True, accumulate="double"):
    """-> (nmatches, match_of_train[n]): -1 untouched, -2 assigned in this call and set to NULL again by the rotation check"""
    T, q = pr["train"], pr["query"]
    n = len(T["x"])
    g = Grid(T["x"], T["y"], T["grid"])
    toct, tang, tur, tdesc = np.asarray(T["octave"]), np.asarray(T["angle"], F32), np.asarray(T["u_right"], F32), np.asarray(T["desc"], np.uint8).reshape(-1, 32)
    blocked = np.asarray(T["occupied"]).astype(bool).copy()
    tcw, tlw = np.asarray(pr["tcw"], F32), np.asarray(pr["tlw"], F32)
    fx, fy, cx, cy, mbf, mb = [F32(v) for v in pr["K6"]]
    minx, maxx, miny, maxy = [F32(v) for v in pr["bounds"]]
    sf = np.asarray(pr["scale_factors"], F32)
    th = F32(pr["th"])
    mono = bool(pr.get("mono"))
    # twc = -Rcw.t() * tcw, tlc = Rlw * twc + tlw  (only the sign and size of tlc[2] against the baseline are used)
    Rcw, t = tcw[:3, :3].astype(np.float64), tcw[:3, 3].astype(np.float64)
    twc = (-(Rcw.T @ t)).astype(F32)                       # (a transposed operand: gemm's general path, double accumulators)
    tlc2 = _mat3_vec_plus(tlw, twc, accumulate)[2]
    forward = bool(tlc2 > mb) and not mono
    backward = bool(-tlc2 > mb) and not mono
    out = np.full(n, -1, np.int64)
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    qxw, qvalid, qoct, qang = np.asarray(q["xw"], F32), np.asarray(q["valid"]), np.asarray(q["octave"]), np.asarray(q["angle"], F32)
    qdesc, qobs = np.asarray(q["desc"], np.uint8).reshape(-1, 32), np.asarray(q["observed"])
    for i in range(len(qvalid)):
        if not qvalid[i]:
            continue
        xc, yc, zc = _mat3_vec_plus(tcw, qxw[i], accumulate)
        invzc = F32(1.0 / float(zc))                       # const float invzc = 1.0 / x3Dc.at<float>(2): double division, float result
        if invzc < 0:
            continue
        u = F32(F32(F32(fx * xc) * invzc) + cx)
        v = F32(F32(F32(fy * yc) * invzc) + cy)
        if u < minx or u > maxx or v < miny or v > maxy:
            continue
        lo = int(qoct[i])
        radius = F32(th * sf[lo])
        if forward:
            cand = g.features_in_area(u, v, radius, toct, lo, -1)
        elif backward:
            cand = g.features_in_area(u, v, radius, toct, 0, lo)
        else:
            cand = g.features_in_area(u, v, radius, toct, lo - 1, lo + 1)
        if not cand:
            continue
        best, bidx = 256, -1
        for j in cand:
            if blocked[j]:
                continue
            if tur[j] > 0:
                ur = F32(u - F32(mbf * invzc))
                if abs(F32(ur - tur[j])) > radius:
                    continue
            d = descriptor_distance(qdesc[i], tdesc[j])
            if d < best:
                best, bidx = d, j
        if best <= TH_HIGH:
            out[bidx] = i
            if qobs[i]:
                blocked[bidx] = True
            nm += 1
            if check_ori:
                hist[_rot_bin(qang[i], tang[bidx])].append(bidx)
    if check_ori:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b in keep:
                continue
            for j in hist[b]:
                out[j] = -2
                nm -= 1
    return nm, out.astype(np.int32)


def search_by_projection_points(pr, nnratio, obj=False):
    """the local-map overload (obj=False) and the object overload (obj=True: fixed 5-pixel window, levels -1 .. +1, bounding-box test,
    threshold 130) -> (nmatches, match_of_train[n])"""
    T, q = pr["train"], pr["query"]
    n = len(T["x"])
    g = Grid(T["x"], T["y"], T["grid"])
    toct, tur, tdesc = np.asarray(T["octave"]), np.asarray(T["u_right"], F32), np.asarray(T["desc"], np.uint8).reshape(-1, 32)
    inbox = np.asarray(T.get("in_bbox", np.ones(n, np.uint8)))
    blocked = np.asarray(T["occupied"]).astype(bool).copy()
    sf = np.asarray(pr["scale_factors"], F32)
    th = F32(pr["th"])
    factor = bool(th != F32(1.0))
    out = np.full(n, -1, np.int64)
    nm = 0
    valid, px, py, pxr = np.asarray(q["valid"]), np.asarray(q["proj_x"], F32), np.asarray(q["proj_y"], F32), np.asarray(q["proj_xr"], F32)
    lvl, vc, qdesc, qobs = np.asarray(q["level"]), np.asarray(q["view_cos"], F32), np.asarray(q["desc"], np.uint8).reshape(-1, 32), np.asarray(q["observed"])
    limit = TH_HIGH_FORDYNAMIC if obj else TH_HIGH
    for i in range(len(valid)):
        if not valid[i]:
            continue
        level = int(lvl[i])
        r = F32(2.5) if float(vc[i]) > 0.998 else F32(4.0)          # the comparison promotes the float to double
        if factor:
            r = F32(r * th)
        if obj:
            cand = g.features_in_area(px[i], py[i], F32(5), toct, level - 1, level + 1)
        else:
            cand = g.features_in_area(px[i], py[i], F32(r * sf[level]), toct, level - 1, level)
        if not cand:
            continue
        best, best2, blevel, blevel2, bidx = 256, 256, -1, -1, -1
        for j in cand:
            if obj and not inbox[j]:
                continue
            if blocked[j]:
                continue
            if tur[j] > 0:
                if abs(F32(pxr[i] - tur[j])) > F32(r * sf[level]):
                    continue
            d = descriptor_distance(qdesc[i], tdesc[j])
            if d < best:
                best2, best, blevel2, blevel, bidx = best, d, blevel, int(toct[j]), j
            elif d < best2:
                blevel2, best2 = int(toct[j]), d
        if best <= limit:
            if blevel == blevel2 and F32(best) > F32(nnratio) * F32(best2):
                continue
            out[bidx] = i
            if qobs[i]:
                blocked[bidx] = True
            nm += 1
    return nm, out.astype(np.int32)
