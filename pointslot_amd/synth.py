"""Seeded synthetic inputs for the BASELINE.json configs.  Pure numpy and
self-contained, so every consumer (GPU path, CPU checker, bench) sees identical bytes.

RNG: xoshiro256** run as LANES independent lock-step streams (each lane seeded by splitmix64 from
(seed, lane)); values are consumed step-major.  Deterministic for a given (seed, lanes).
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return x, z ^ (z >> np.uint64(31))


def _rotl(x, k):
    return ((x << np.uint64(k)) | (x >> np.uint64(64 - k))) & _M


class Rng:
    def __init__(self, seed, lanes=256):
        with np.errstate(over="ignore"):
            x = (np.uint64(seed) + np.arange(lanes, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)) & _M
            s = []
            for _ in range(4):
                x, z = _splitmix64(x)
                s.append(z)
        self.s = s
        self.lanes = lanes
        self._buf = np.zeros(0, np.uint64)

    def _step(self):
        with np.errstate(over="ignore"):
            s0, s1, s2, s3 = self.s
            r = (_rotl((s1 * np.uint64(5)) & _M, 7) * np.uint64(9)) & _M
            t = (s1 << np.uint64(17)) & _M
            s2 = s2 ^ s0
            s3 = s3 ^ s1
            s1 = s1 ^ s2
            s0 = s0 ^ s3
            s2 = s2 ^ t
            s3 = _rotl(s3, 45)
            self.s = [s0, s1, s2, s3]
        return r

    def u64(self, n):
        while self._buf.size < n:
            self._buf = np.concatenate([self._buf, self._step()])
        out, self._buf = self._buf[:n], self._buf[n:]
        return out

    def uniform(self, n, lo=0.0, hi=1.0):
        u = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def integers(self, n, lo, hi):
        """uniform integers in [lo, hi)"""
        return (lo + np.floor(self.uniform(n) * (hi - lo))).astype(np.int64)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return z[:n]


# ---- config 2: stereo pair for the ORB extractor ---------------------------------------------------
def _value_noise(rng, w, h, cell, amp):
    gw, gh = w // cell + 2, h // cell + 2
    g = rng.uniform(gw * gh, -1.0, 1.0).reshape(gh, gw)
    ys, xs = np.arange(h) / cell, np.arange(w) / cell
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def stereo_pair(seed=0x51070002, w=1242, h=375, n_rect=400, bf=384.38148):
    """Value-noise texture (amplitude 64/32/16 on lattices of 32/16/8 px, plus a fine 3-px octave of
    amplitude 14 so that FAST sees a realistic few thousand candidates per level) plus `n_rect`
    uniform grey rectangles of 8..64 px for corners; right image = left warped by the disparity
    d(y) = bf / z(y), z linear 6 m (bottom) .. 60 m (top), bilinear.  Returns two uint8 (h, w) arrays."""
    rng = Rng(seed)
    img = np.full((h, w), 128.0)
    for cell, amp in ((32, 64.0), (16, 32.0), (8, 16.0), (3, 14.0)):
        img += _value_noise(rng, w, h, cell, amp)
    rw = rng.integers(n_rect, 8, 65); rh = rng.integers(n_rect, 8, 65)
    rx = rng.integers(n_rect, 0, w); ry = rng.integers(n_rect, 0, h)
    rg = rng.integers(n_rect, 0, 256)
    for i in range(n_rect):
        img[ry[i]:ry[i] + rh[i], rx[i]:rx[i] + rw[i]] = rg[i]
    left = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    z = 60.0 + (6.0 - 60.0) * (np.arange(h) / (h - 1))
    d = bf / z
    xs = np.arange(w)[None, :] + d[:, None]          # right(x) = left(x + d)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    fx = xs - np.floor(xs)
    rows = np.arange(h)[:, None]
    lf = left.astype(np.float64)
    right = np.clip(np.rint(lf[rows, x0] * (1 - fx) + lf[rows, x1] * fx), 0, 255).astype(np.uint8)
    return left, right


def stereo_batch(n_pairs, seed=0x51070002, w=1242, h=375):
    """`n_pairs` stereo pairs (seed + k) as one uint8 array [2 * n_pairs, h, w] (L0, R0, L1, R1, ...)."""
    out = np.zeros((2 * n_pairs, h, w), np.uint8)
    for k in range(n_pairs):
        l, r = stereo_pair(seed + k, w, h)
        out[2 * k], out[2 * k + 1] = l, r
    return out
