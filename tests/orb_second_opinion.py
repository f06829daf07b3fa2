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
