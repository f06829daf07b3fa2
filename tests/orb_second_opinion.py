"""A second, independently written statement of the pixel stages of the ORB extractor (numpy, brute-force definitions) used
ONLY to cross-check oracle/orb_oracle.cpp in the CPU tests.  The oracle restates OpenCV's code paths
(scan-line FAST with its score buffers, cornerScore<16>, the resize row/column tables ...); this file states WHAT those paths
compute, from the definitions, sharing no code with it:

  resize      dst = bilinear with the 11-bit fixed-point coefficients of cv::resize(INTER_LINEAR) on 8-bit data
  border      BORDER_REFLECT_101 == numpy.pad(mode="reflect")
  blur        separable 7 x 7, sigma 2, 8.8 fixed-point kernel, one rounding (x + 2^15) >> 16
  FAST        a pixel is a corner at threshold t iff 9 contiguous ring pixels are all > v + t or all < v - t; its score is the
              largest t for which that holds (found by bisection over t on the boolean definition); 3 x 3 strict non-maximum
              suppression inside the cell's candidate area; iniThFAST, then minThFAST when the cell is empty
  quadtree    a list-based transcription of ORBextractor::DistributeOctTree / ExtractorNode::DivideNode
              (/root/reference/src/ORBextractor.cc:481-763)

It pins nothing against OpenCV itself (neither does the oracle: OpenCV is not available in the build image); it removes the
single-author risk of the oracle's C++ the way the dense numpy LM does for the optimisers."""
import math

import numpy as np

EDGE = 19
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]   # (dx, dy)


def cv_round(x):
    return int(np.rint(x))           # round half to even, like cvRound


def scale_tables(nlevels=8, scale=1.2):
    sf = [np.float32(1.0)]
    for _ in range(1, nlevels):
        sf.append(np.float32(np.float64(sf[-1]) * np.float64(np.float32(scale))))   # mvScaleFactor[i] = mvScaleFactor[i-1] * scaleFactor (double member)
    return np.array(sf, np.float32), (np.float32(1.0) / np.array(sf, np.float32)).astype(np.float32)


def _coefficients(n_dst, n_src, clamp_fraction):
    """cv::resize(INTER_LINEAR), 8-bit: source index and the two 11-bit weights of every destination coordinate"""
    scale = 1.0 / (n_dst / n_src)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_fraction:               # columns: outside the image the nearest pixel alone is used
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0
        s[hi] = n_src - 1
    w1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
    w0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
    return s, w0, w1


def resize_linear_u8(src, dw, dh):
    sh, sw = src.shape
    sx, a0, a1 = _coefficients(dw, sw, True)
    sy, b0, b1 = _coefficients(dh, sh, False)
    S = src.astype(np.int64)
    rows = S[:, sx] * a0[None, :] + S[:, np.minimum(sx + 1, sw - 1)] * a1[None, :]        # horizontal pass, exact integers
    r0 = rows[np.clip(sy, 0, sh - 1)]
    r1 = rows[np.clip(sy + 1, 0, sh - 1)]
    out = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def pyramid(img, nlevels=8, scale=1.2):
    """level images (ORBextractor::ComputePyramid, ORBextractor.cc:1107-1132): level l is resized from level l - 1"""
    _, inv = scale_tables(nlevels, scale)
    h, w = img.shape
    levels = [np.ascontiguousarray(img)]
    for l in range(1, nlevels):
        wl, hl = cv_round(np.float32(w) * inv[l]), cv_round(np.float32(h) * inv[l])
        levels.append(resize_linear_u8(levels[-1], wl, hl))
    return levels


def padded(level_img):
    return np.pad(level_img, EDGE, mode="reflect")


def gaussian_blur7(img):
    x = np.arange(7) - 3.0
    g = np.exp(-0.5 * x * x / 4.0)
    k = np.rint(g / g.sum() * 256.0).astype(np.int64)            # 8.8 fixed point
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    h, w = img.shape
    hor = sum(k[i] * p[:, i:i + w] for i in range(7))
    hor = np.minimum(hor, 65535)
    ver = sum(k[j] * hor[j:j + h, :] for j in range(7))
    return np.minimum((ver + 32768) >> 16, 255).astype(np.uint8)


def fast_score_map(img):
    """score[y, x] = the largest t >= 0 such that 9 contiguous ring pixels are all > v + t or all < v - t; -1 where the pixel is
    not a corner even at t = 0.  Defined for pixels at least 3 away from the border (-1 elsewhere)."""
    h, w = img.shape
    I = img.astype(np.int32)
    v = I[3:h - 3, 3:w - 3]
    D = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - v for dx, dy in RING])        # [16, h-6, w-6]

    def is_corner(t):
        out = np.zeros(v.shape, bool)
        for sign in (1, -1):
            m = np.zeros(v.shape, np.uint32)
            for i in range(16):
                m |= ((sign * D[i] > t).astype(np.uint32)) << np.uint32(i)
            m |= m << np.uint32(16)                                  # the ring is circular
            run = m.copy()
            for s in range(1, 9):
                run &= m >> np.uint32(s)
            out |= (run & np.uint32(0xFFFF)) != 0
        return out

    lo = np.full(v.shape, -1, np.int32)      # largest t known to be a corner
    hi = np.full(v.shape, 255, np.int32)     # smallest t known not to be one (no difference exceeds 255)
    while np.any(hi - lo > 1):
        mid = (lo + hi) // 2
        c = is_corner(mid)
        lo = np.where(c, mid, lo)
        hi = np.where(c, hi, mid)
    out = np.full((h, w), -1, np.int32)
    out[3:h - 3, 3:w - 3] = lo
    return out


def fast_cells(level_img, ini_th=20, min_th=5):
    """candidates of one level in the reference's emission order, (x, y, score) relative to (minBorderX, minBorderY)
    (ORBextractor::ComputeKeyPointsOctTree, ORBextractor.cc:765-829; cv::FAST with non-maximum suppression on every cell)"""
    h, w = level_img.shape
    score = fast_score_map(level_img)
    min_b = EDGE - 3
    max_bx, max_by = w - EDGE + 3, h - EDGE + 3
    width, height = np.float32(max_bx - min_b), np.float32(max_by - min_b)
    n_cols, n_rows = int(width / np.float32(30)), int(height / np.float32(30))
    w_cell, h_cell = int(math.ceil(width / np.float32(n_cols))), int(math.ceil(height / np.float32(n_rows)))
    out = []
    for i in range(n_rows):
        y0 = min_b + i * h_cell
        y1 = min(y0 + h_cell + 6, max_by)
        if y0 >= max_by - 3:
            continue
        for j in range(n_cols):
            x0 = min_b + j * w_cell
            x1 = min(x0 + w_cell + 6, max_bx)
            if x0 >= max_bx - 3:
                continue
            roi = score[y0:y1, x0:x1]
            rh, rw = roi.shape
            if rh < 7 or rw < 7:
                continue
            for th in (ini_th, min_th):
                s = np.zeros((rh, rw), np.int32)                     # scores of the corners at th inside the candidate area, else 0
                inner = roi[3:rh - 3, 3:rw - 3]
                s[3:rh - 3, 3:rw - 3] = np.where(inner >= th, inner, 0)
                c = s[3:rh - 3, 3:rw - 3]
                keep = c > 0
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        if dx or dy:
                            keep &= c > s[3 + dy:rh - 3 + dy, 3 + dx:rw - 3 + dx]
                ys, xs = np.nonzero(keep)                            # raster order
                if len(ys):
                    for y, x in zip(ys, xs):
                        out.append((x + 3 + j * w_cell, y + 3 + i * h_cell, int(c[y, x])))
                    break
    return np.array(out, np.int32).reshape(-1, 3)


class _Node:
    __slots__ = ("ul", "ur", "bl", "br", "keys", "no_more", "seq")

    def __init__(self):
        self.keys = []
        self.no_more = False
        self.seq = 0


def _divide(n):
    half_x = int(math.ceil(np.float32(n.ur[0] - n.ul[0]) / 2))
    half_y = int(math.ceil(np.float32(n.br[1] - n.ul[1]) / 2))
    c = [_Node() for _ in range(4)]
    c[0].ul = n.ul; c[0].ur = (n.ul[0] + half_x, n.ul[1]); c[0].bl = (n.ul[0], n.ul[1] + half_y); c[0].br = (n.ul[0] + half_x, n.ul[1] + half_y)
    c[1].ul = c[0].ur; c[1].ur = n.ur; c[1].bl = c[0].br; c[1].br = (n.ur[0], n.ul[1] + half_y)
    c[2].ul = c[0].bl; c[2].ur = c[0].br; c[2].bl = n.bl; c[2].br = (c[0].br[0], n.bl[1])
    c[3].ul = c[2].ur; c[3].ur = c[1].br; c[3].bl = c[2].br; c[3].br = n.br
    for k in n.keys:
        if k[0] < c[0].ur[0]:
            (c[0] if k[1] < c[0].br[1] else c[2]).keys.append(k)
        elif k[1] < c[0].br[1]:
            c[1].keys.append(k)
        else:
            c[3].keys.append(k)
    for ch in c:
        if len(ch.keys) == 1:
            ch.no_more = True
    return c


def distribute(keys, min_x, max_x, min_y, max_y, n_wanted):
    """ORBextractor::DistributeOctTree.  keys: (x, y, response) rows relative to (minX, minY).  The reference sorts
    pair<int, ExtractorNode*>: ties on the count fall to heap addresses; here, as in the oracle and the GPU kernel, to the node's
    creation order (a later node compares larger)."""
    n_ini = int(math.floor(float(np.float32(max_x - min_x) / np.float32(max_y - min_y)) + 0.5))      # C round(): half away from zero
    h_x = np.float32(max_x - min_x) / np.float32(n_ini)
    counter = [0]

    def stamp(node):
        counter[0] += 1
        node.seq = counter[0]
        return node

    nodes = []
    for i in range(n_ini):
        n = stamp(_Node())
        n.ul = (int(h_x * np.float32(i)), 0); n.ur = (int(h_x * np.float32(i + 1)), 0)
        n.bl = (n.ul[0], max_y - min_y); n.br = (n.ur[0], max_y - min_y)
        nodes.append(n)
    for k in keys:
        nodes[int(np.float32(k[0]) / h_x)].keys.append((int(k[0]), int(k[1]), int(k[2])))
    ini = nodes
    nodes = []
    for n in ini:
        if len(n.keys) == 1:
            n.no_more = True
        if n.keys:
            nodes.append(n)
    finish = False
    while not finish:
        prev_size = len(nodes)
        to_expand = 0
        expandable = []
        i = 0
        while i < len(nodes):
            n = nodes[i]
            if n.no_more:
                i += 1
                continue
            for ch in _divide(n):
                if ch.keys:
                    nodes.insert(0, stamp(ch))            # push_front
                    i += 1
                    if len(ch.keys) > 1:
                        to_expand += 1
                        expandable.append(ch)
            del nodes[i]                                   # lit = lNodes.erase(lit)
        if len(nodes) >= n_wanted or len(nodes) == prev_size:
            finish = True
        elif len(nodes) + to_expand * 3 > n_wanted:
            while not finish:
                prev_size = len(nodes)
                prev = sorted(expandable, key=lambda nd: (len(nd.keys), nd.seq))
                expandable = []
                for nd in reversed(prev):
                    for ch in _divide(nd):
                        if ch.keys:
                            nodes.insert(0, stamp(ch))
                            if len(ch.keys) > 1:
                                expandable.append(ch)
                    nodes.remove(nd)
                    if len(nodes) >= n_wanted:
                        break
                if len(nodes) >= n_wanted or len(nodes) == prev_size:
                    finish = True
    out = []
    for n in nodes:
        best = n.keys[0]
        for k in n.keys[1:]:
            if k[2] > best[2]:
                best = k
        out.append(best)
    return np.array(out, np.int32).reshape(-1, 3)


# ---- cv::ORB (the object-feature detector): the stages that differ from ORBextractor ------------------------
def _exact_coefficients(n_dst, n_src):
    """resize(INTER_LINEAR_EXACT), 8-bit: source offset, 8.8 weights, and the range of destination indices with two source samples"""
    scale = 1.0 / (n_dst / n_src)
    f = scale * (np.arange(n_dst, dtype=np.float64) + 0.5) - 0.5
    i = np.floor(f).astype(np.int64)
    c1 = np.rint((f - i) * 256.0).astype(np.int64)
    before = i < 0
    after = i >= n_src - 1
    return np.clip(i, 0, n_src - 2), 256 - c1, c1, before, after


def resize_linear_exact_u8(src, dw, dh):
    sh, sw = src.shape
    xo, a0, a1, xb, xa = _exact_coefficients(dw, sw)
    yo, b0, b1, yb, ya = _exact_coefficients(dh, sh)
    S = src.astype(np.int64)
    hor = S[:, xo] * a0 + S[:, xo + 1] * a1
    hor[:, xb] = S[:, :1] << 8
    hor[:, xa] = S[:, -1:] << 8
    out = (hor[yo] * b0[:, None] + hor[yo + 1] * b1[:, None] + 32768) >> 16
    out[yb] = (hor[0] + 128) >> 8
    out[ya] = (hor[-1] + 128) >> 8
    return np.clip(out, 0, 255).astype(np.uint8)


def cv_pyramid(img, nlevels=8, scale=1.2):
    """cv::ORB's pyramid: level l = INTER_LINEAR_EXACT resize of level l - 1 to cvRound(size / (float)pow(scaleFactor, l))"""
    sf = np.float64(np.float32(scale))
    h, w = img.shape
    levels = [np.ascontiguousarray(img)]
    for l in range(1, nlevels):
        inv = np.float32(1.0) / np.float32(sf ** l)
        levels.append(resize_linear_exact_u8(levels[-1], cv_round(np.float32(w) * inv), cv_round(np.float32(h) * inv)))
    return levels


def cv_fast(level_img, threshold=20, edge=19, mask=None):
    """cv::FAST(threshold, nonmaxSuppression) on the whole level, then runByPixelsMask and runByImageBorder(edge): (x, y, score)"""
    s = fast_score_map(level_img)
    h, w = s.shape
    c = np.where(s >= threshold, s, 0)
    p = np.pad(c, 1)
    keep = c > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                keep &= c > p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    if mask is not None:
        keep &= mask != 0
    inner = np.zeros_like(keep)
    inner[edge:h - edge, edge:w - edge] = True
    ys, xs = np.nonzero(keep & inner)
    return np.stack([xs, ys, c[ys, xs]], 1).astype(np.int32)


def cv_harris(level_img, xs, ys):
    """HarrisResponses(blockSize 7, k 0.04) as float32 arithmetic in the order orb.cpp writes it"""
    I = np.pad(level_img.astype(np.int64), 8, mode="reflect")
    out = np.zeros(len(xs), np.float32)
    scale = np.float32(1.0) / (np.float32(28) * np.float32(255.0))
    s4 = scale * scale * scale * scale
    for n, (x, y) in enumerate(zip(xs, ys)):
        blk = I[y + 8 - 4:y + 8 + 5, x + 8 - 4:x + 8 + 5]        # 9 x 9 around the point
        Ix = (blk[1:-1, 2:] - blk[1:-1, :-2]) * 2 + (blk[:-2, 2:] - blk[:-2, :-2]) + (blk[2:, 2:] - blk[2:, :-2])
        Iy = (blk[2:, 1:-1] - blk[:-2, 1:-1]) * 2 + (blk[2:, :-2] - blk[:-2, :-2]) + (blk[2:, 2:] - blk[:-2, 2:])
        a, b, c = np.float32(int((Ix * Ix).sum())), np.float32(int((Iy * Iy).sum())), np.float32(int((Ix * Iy).sum()))
        out[n] = (a * b - c * c - np.float32(0.04) * (a + b) * (a + b)) * s4
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
# r04: the stages behind the keypoint selection, again from the definitions: IC_Angle + fastAtan2, the steered BRIEF
# comparisons, Frame::ComputeStereoMatches, cv::RNG.
# ------------------------------------------------------------------------------------------------------------------------------------
HALF_PATCH = 15


def umax_table():
    """ORBextractor.cc:452-470: the half-widths of the 31-pixel disc's rows, from the circle equation and its mirror-symmetric completion"""
    umax = np.zeros(HALF_PATCH + 1, np.int64)
    vmax = int(np.floor(HALF_PATCH * np.sqrt(2.0) / 2 + 1))
    vmin = int(np.ceil(HALF_PATCH * np.sqrt(2.0) / 2))
    hp2 = HALF_PATCH * HALF_PATCH
    for v in range(vmax + 1):
        umax[v] = int(cv_round(np.sqrt(float(hp2 - v * v))))
    v0 = 0
    for v in range(HALF_PATCH, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    return umax


def fast_atan2_deg(y, x):
    """OpenCV's scalar fastAtan2 (degrees) in IEEE float32, operation by operation: a 7th-order odd polynomial in min / max of |x|, |y|
    (+ DBL_EPSILON rounded to float in the denominator), then the octant and quadrant reflections"""
    f = np.float32
    p1 = f(0.9997878412794807) * f(180.0 / np.pi)
    p3 = f(-0.3258083974640975) * f(180.0 / np.pi)
    p5 = f(0.1555786518463281) * f(180.0 / np.pi)
    p7 = f(-0.04432655554792128) * f(180.0 / np.pi)
    y = np.asarray(y, f); x = np.asarray(x, f)
    ax, ay = np.abs(x), np.abs(y)
    eps = f(2.220446049250313e-16)
    big, small = np.maximum(ax, ay), np.minimum(ax, ay)
    with np.errstate(invalid="ignore", divide="ignore"):
        c = (small / (big + eps)).astype(f)
    c2 = (c * c).astype(f)
    a = ((((((p7 * c2).astype(f) + p5).astype(f) * c2).astype(f) + p3).astype(f) * c2).astype(f) + p1).astype(f)
    a = (a * c).astype(f)
    a = np.where(ax >= ay, a, (f(90.0) - a).astype(f)).astype(f)
    a = np.where(x < 0, (f(180.0) - a).astype(f), a).astype(f)
    a = np.where(y < 0, (f(360.0) - a).astype(f), a).astype(f)
    return a


def ic_angles(padded_plane, xs, ys, border=EDGE):
    """IC_Angle (ORBextractor.cc:77-104) as two masked sums over the 31 x 31 neighbourhood: m10 = sum u I, m01 = sum v I over the disc"""
    um = umax_table()
    v, u = np.mgrid[-HALF_PATCH:HALF_PATCH + 1, -HALF_PATCH:HALF_PATCH + 1]
    disc = np.abs(u) <= um[np.abs(v)]
    out = np.zeros(len(xs), np.float32)
    m01s = np.zeros(len(xs), np.int64); m10s = np.zeros(len(xs), np.int64)
    for i, (x, y) in enumerate(zip(xs, ys)):
        cx, cy = int(cv_round(float(x))) + border, int(cv_round(float(y))) + border
        patch = padded_plane[cy - HALF_PATCH:cy + HALF_PATCH + 1, cx - HALF_PATCH:cx + HALF_PATCH + 1].astype(np.int64)
        m10s[i] = int((patch * u)[disc].sum()); m01s[i] = int((patch * v)[disc].sum())
    out = fast_atan2_deg(m01s.astype(np.float32), m10s.astype(np.float32))
    return out


def steered_brief(blurred_level, xs, ys, angles_deg, pattern):
    """computeOrbDescriptor (ORBextractor.cc:108-147): the 256 comparisons of the pattern rotated by the keypoint angle - cos / sin of
    the float angle in radians, every product and sum rounded to float32, cvRound = round-half-to-even"""
    f = np.float32
    pat = np.asarray(pattern, np.int64).reshape(256, 2, 2)            # [bit][tap 0 / 1][x, y]
    out = np.zeros((len(xs), 32), np.uint8)
    factor = f(np.pi / 180.0)
    for i, (x, y, ang) in enumerate(zip(xs, ys, angles_deg)):
        rad = f(f(ang) * factor)
        a, b = f(np.cos(np.float64(rad))), f(np.sin(np.float64(rad)))
        px = pat[:, :, 0].astype(f); py = pat[:, :, 1].astype(f)
        yy = np.rint(((px * b).astype(f) + (py * a).astype(f)).astype(f)).astype(np.int64)
        xx = np.rint(((px * a).astype(f) - (py * b).astype(f)).astype(f)).astype(np.int64)
        cx, cy = int(cv_round(float(x))), int(cv_round(float(y)))
        vals = blurred_level[cy + yy, cx + xx]                          # [256][2]
        bits = (vals[:, 0] < vals[:, 1]).astype(np.uint8)
        out[i] = np.packbits(bits.reshape(32, 8), axis=1, bitorder="little")[:, 0]
    return out


def _popcount_rows(a, b):
    return np.unpackbits(a ^ b, axis=-1).sum(-1)


def stereo_matches(kps_l, desc_l, kps_r, desc_r, pyr_l, pyr_r, scale, inv_scale, mb, mbf):
    """Frame::ComputeStereoMatches (/root/reference/src/Frame.cc:2142-2316) read off the reference: the row table of the right keypoints,
    best Hamming distance inside the disparity range and one octave, the 11 x 11 L1 slide over +-5 columns with `int bestDist` receiving
    the float SAD, the parabola, the disparity clamp (in DOUBLE where the reference subtracts the literal 0.01), the 1.5f * 1.4f * median cut.
    kps: structured arrays with x, y, octave.  pyr: the (unpadded) level images."""
    f = np.float32
    n = len(kps_l)
    u_right = np.full(n, -1.0, f); depth = np.full(n, -1.0, f)
    n_rows = pyr_l[0].shape[0]
    rows = [[] for _ in range(n_rows)]
    for ir in range(len(kps_r)):
        ky = f(kps_r["y"][ir]); r = f(f(2.0) * f(scale[int(kps_r["octave"][ir])]))
        for yi in range(int(np.floor(f(ky - r))), int(np.ceil(f(ky + r))) + 1):
            if 0 <= yi < n_rows:
                rows[yi].append(ir)
    min_z, min_d = f(mb), f(0)
    max_d = f(f(mbf) / min_z)
    th_orb = (100 + 50) // 2
    dist_idx = []
    xr = kps_r["x"].astype(f); octr = kps_r["octave"].astype(np.int64)
    for il in range(n):
        ul, vl, lev = f(kps_l["x"][il]), f(kps_l["y"][il]), int(kps_l["octave"][il])
        cand = np.asarray(rows[int(vl)], np.int64)
        if len(cand) == 0:
            continue
        min_u, max_u = f(ul - max_d), f(ul - min_d)
        if max_u < 0:
            continue
        ok = (octr[cand] >= lev - 1) & (octr[cand] <= lev + 1) & (xr[cand] >= min_u) & (xr[cand] <= max_u)
        cand = cand[ok]
        best_dist, best_r = 100, 0
        if len(cand):
            d = _popcount_rows(desc_r[cand], desc_l[il][None, :])
            k = int(np.argmin(d))                                       # the first smallest: `dist < bestDist` keeps the earliest
            if d[k] < best_dist:
                best_dist, best_r = int(d[k]), int(cand[k])
        if best_dist >= th_orb:
            continue
        ur0 = xr[best_r]
        sf = f(inv_scale[lev])
        su_l = f(np.floor(abs(f(ul * sf)) + 0.5) * np.sign(f(ul * sf)))   # round(): half away from zero
        sv_l = f(np.floor(abs(f(vl * sf)) + 0.5) * np.sign(f(vl * sf)))
        su_r0 = f(np.floor(abs(f(ur0 * sf)) + 0.5) * np.sign(f(ur0 * sf)))
        w = L = 5
        il_img, ir_img = pyr_l[lev], pyr_r[lev]
        cy, cxl = int(sv_l), int(su_l)
        iniu, endu = f(su_r0 + L - w), f(su_r0 + L + w + 1)
        if iniu < 0 or endu >= ir_img.shape[1]:
            continue
        pl = il_img[cy - w:cy + w + 1, cxl - w:cxl + w + 1].astype(f)
        pl = pl - pl[w, w]
        best_s, best_inc = 2 ** 31 - 1, 0
        dists = np.zeros(2 * L + 1, f)
        for inc in range(-L, L + 1):
            cxr = int(f(su_r0 + inc))
            pr = ir_img[cy - w:cy + w + 1, cxr - w:cxr + w + 1].astype(f)
            pr = pr - pr[w, w]
            dist = f(np.abs(pl - pr).astype(np.float64).sum())           # cv::norm accumulates in double; the values are integers
            if dist < best_s:
                best_s, best_inc = int(dist), inc
            dists[L + inc] = dist
        if best_inc == -L or best_inc == L:
            continue
        d1, d2, d3 = dists[L + best_inc - 1], dists[L + best_inc], dists[L + best_inc + 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = f(f(d1 - d3) / f(f(2.0) * f(f(d1 + d3) - f(f(2.0) * d2))))
        if not (delta >= -1 and delta <= 1):                              # (a NaN parabola passes `deltaR<-1 || deltaR>1` in C++ ...)
            if not np.isnan(delta):
                continue
        best_ur = f(f(scale[lev]) * f(f(su_r0 + f(best_inc)) + delta))
        disparity = f(ul - best_ur)
        if disparity >= min_d and disparity < max_d:                      # (... and fails here)
            if disparity <= 0:
                disparity = f(0.01)
                best_ur = f(np.float64(ul) - 0.01)
            depth[il] = f(f(mbf) / disparity); u_right[il] = best_ur
            dist_idx.append((best_s, il))
    if not dist_idx:
        return 0, u_right, depth
    dist_idx.sort()
    median = f(dist_idx[len(dist_idx) // 2][0])
    th = f(f(f(1.5) * f(1.4)) * median)
    kept = len(dist_idx)
    for s, il in reversed(dist_idx):
        if s < th:
            break
        u_right[il] = -1; depth[il] = -1; kept -= 1
    return kept, u_right, depth


def cv_rng_closed_form(n_draws, state0=0xFFFFFFFF):
    """cv::RNG's multiply-with-carry stream WITHOUT iterating its update rule: with b = 2^32 and a = 4164903690 the 64-bit state
    s = carry * b + x satisfies b * s' = s (mod a b - 1), so s_k = s_0 * b^-k mod (a b - 1); the k-th output is s_k mod 2^32."""
    a, b = 4164903690, 1 << 32
    m = a * b - 1
    binv = pow(b, -1, m)
    out = np.zeros(n_draws, np.uint64)
    s = state0
    for k in range(n_draws):
        s = (s * binv) % m
        out[k] = s & 0xFFFFFFFF
    return out
