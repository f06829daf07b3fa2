"""Second opinions for the SURVEY 8f-4 rows (CPU only, numpy from the reference's text, no code shared with oracle/):
  MapObjectPoint / MapPoint::ComputeDistinctiveDescriptors   /root/reference/src/MapObjectPoint.cc:379-436
  the reprojection test of Tracking::DynamicStaticDiscrimination   /root/reference/src/Tracking.cc:2099-2181"""
import numpy as np

import opt_second_opinion as so
import oracle_lib
from golden_cases import distinctive_case, dynamic_cases
from pointslot_amd import synth

_POP = np.array([bin(i).count("1") for i in range(256)], np.int64)


def _distinctive(desc):
    """the observation with the least median distance to the others: int median = vDists[0.5 * (N - 1)] of every SORTED row, first minimum wins"""
    d = np.asarray(desc, np.uint8).reshape(-1, 32)
    n = len(d)
    D = _POP[d[:, None, :] ^ d[None, :, :]].sum(2)
    best, bidx = 2 ** 31 - 1, 0
    for i in range(n):
        med = int(np.sort(D[i])[int(0.5 * (n - 1))])
        if med < best:
            best, bidx = med, i
    return bidx


def test_distinctive_descriptors_equal_the_restatement():
    lists = distinctive_case()
    rng = np.random.default_rng(5)
    for n in (2, 4, 9, 33, 100):                       # ties between rows: the FIRST least median is kept
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        lists.append(np.array([base if k % 3 else ~base for k in range(n)], np.uint8))
    got = oracle_lib.distinctive_descriptors(lists)
    assert [int(g) for g in got] == [_distinctive(l) for l in lists]


def _pose(p7):
    p7 = np.asarray(p7, float)
    return so.normalize_rotation(p7[3:7].copy()), p7[:3].copy()


def _inverse(p):
    q = np.array([-p[0][0], -p[0][1], -p[0][2], p[0][3]])
    return q, so.quat_rotate(q, p[1] * -1.0)


def _dsd(o):
    """-> (mono average, stereo average, mono count, stereo count) as the reference leaves them: below five values of a kind no average is formed"""
    fx, fy, cx, cy = [float(v) for v in o["K"]]
    mbf = float(np.float32(o["mbf"]))
    rel = so.se3_mul(_pose(o["cur_tcw"]), _inverse(_pose(o["last_tcw"])))          # current_pose * last_pose.inverse()
    tco = _pose(o["last_tco"])
    vals = {1: [], 2: []}
    for j in np.nonzero(np.asarray(o["valid"]))[0]:
        plc = so.quat_rotate(tco[0], np.asarray(o["po"][j], float)) + tco[1]
        pc = so.quat_rotate(rel[0], plc) + rel[1]
        invz = 1.0 / pc[2]
        w = float(np.float32(o["inv_sigma2"][j]))
        ob = np.asarray(o["obs"][j], np.float32).astype(np.float64)
        z0, z1 = cx + pc[0] * invz * fx, cy + pc[1] * invz * fy
        e = [ob[0] - z0, ob[1] - z1]
        if ob[2] < 0:
            vals[1].append(sum(x * (w * x) for x in e))
        else:
            e.append(ob[2] - (z0 - mbf * invz))
            vals[2].append(sum(x * (w * x) for x in e))
    out = []
    for kind in (1, 2):
        v = sorted(vals[kind])
        n = len(v)
        avg = 0.0
        if n >= 5:
            med = v[int(len(v) // 2 + 0.5)]
            v = [x for x in v if not x > 5 * med]        # erased while walking the sorted vector: what stays is a prefix
            n = len(v)
            acc = 0.0
            for x in v:
                acc += x
            avg = acc / n
        out.append((avg, n))
    return out[0][0], out[1][0], out[0][1], out[1][1]


def test_dynamic_static_reprojection_test_equals_the_restatement():
    cases = list(dynamic_cases()) + [synth.dynamic_object(300 + k, n=n, moving=m, mono_frac=f, valid_frac=v)
                                     for k, (n, m, f, v) in enumerate(((4, 0.3, 0.0, 1.0), (12, 0.0, 0.5, 0.5), (500, 1.0, 0.2, 0.9), (60, 0.05, 1.0, 1.0)))]
    for k, o in enumerate(cases):
        am, as_, nm, ns = _dsd(o)
        bm, bs, cm, cs = oracle_lib.dynamic_discrimination(o)
        assert (nm, ns) == (cm, cs), (k, nm, ns, cm, cs)
        assert abs(am - bm) <= 1e-9 * max(1.0, abs(bm)) and abs(as_ - bs) <= 1e-9 * max(1.0, abs(bs)), (k, am, bm, as_, bs)


# ---- the search half of ORBmatcher::Fuse(ObjectKeyFrame*, points, th)   /root/reference/src/ORBmatcher.cc:1138-1260 -----------------------------
def _fuse(pr, accumulate="double"):
    """-> (best_idx, best_dist) per candidate point.  Every float of the reference a numpy float32; `Rco * Poj + tco` by either reading of
    the cv::Mat expression (tests/match_second_opinion.py); cv::norm and Mat::dot accumulate in double; GetMin / MaxDistanceInvariance apply 0.8f / 1.2f."""
    import math
    import match_second_opinion as ms
    F32 = np.float32
    T, q = pr["train"], pr["query"]
    g = ms.Grid(T["x"], T["y"], T["grid"])
    tx, ty, toct, tur, tdesc = np.asarray(T["x"], F32), np.asarray(T["y"], F32), np.asarray(T["octave"]), np.asarray(T["u_right"], F32), np.asarray(T["desc"], np.uint8).reshape(-1, 32)
    fx, fy, cx, cy, bf = [F32(v) for v in pr["K5"]]
    R, t, ow = np.asarray(pr["R"], F32), np.asarray(pr["t"], F32), np.asarray(pr["ow"], F32)
    Tm = np.zeros((3, 4), F32); Tm[:, :3] = R; Tm[:, 3] = t
    b = [float(v) for v in pr["bounds"]]
    sf, is2 = np.asarray(pr["scale_factors"], F32), np.asarray(pr["inv_level_sigma2"], F32)
    logsf, nlev, th = F32(pr["log_scale_factor"]), int(pr["n_levels"]), F32(pr["th"])
    m = len(q["valid"])
    bi, bd = np.full(m, -1, np.int32), np.full(m, 256, np.int32)
    for i in range(m):
        if not q["valid"][i]:
            continue
        P = np.asarray(q["pos"][i], F32)
        pc = ms._mat3_vec_plus(Tm, P, accumulate)
        if pc[2] < 0:
            continue
        invz = F32(F32(1) / pc[2])
        x, y = F32(pc[0] * invz), F32(pc[1] * invz)
        u, v = F32(F32(fx * x) + cx), F32(F32(fy * y) + cy)
        ur = F32(u - F32(bf * invz))
        if not (float(u) >= b[0] and float(u) < b[1] and float(v) >= b[2] and float(v) < b[3]):
            continue
        po = [F32(P[c] - ow[c]) for c in range(3)]
        dist = F32(math.sqrt(sum(float(c) * float(c) for c in po)))
        if dist < F32(F32(0.8) * F32(q["min_dist"][i])) or dist > F32(F32(1.2) * F32(q["max_dist"][i])):
            continue
        n = np.asarray(q["normal"][i], F32)
        if sum(float(po[c]) * float(n[c]) for c in range(3)) < 0.5 * float(dist):
            continue
        ratio = F32(F32(q["max_dist"][i]) / dist)
        lvl = int(math.ceil(math.log(float(ratio)) / float(logsf)))
        lvl = 0 if lvl < 0 else min(lvl, nlev - 1)
        radius = F32(th * sf[lvl])
        best, bidx = 256, -1
        for j in g.features_in_area(u, v, radius, toct):
            kl = int(toct[j])
            if kl < lvl - 1 or kl > lvl:
                continue
            ex, ey = F32(u - tx[j]), F32(v - ty[j])
            if tur[j] >= 0:
                er = F32(ur - tur[j])
                e2 = F32(F32(F32(ex * ex) + F32(ey * ey)) + F32(er * er))
                if float(F32(e2 * is2[kl])) > 7.8:
                    continue
            else:
                e2 = F32(F32(ex * ex) + F32(ey * ey))
                if float(F32(e2 * is2[kl])) > 5.99:
                    continue
            d = ms.descriptor_distance(np.asarray(q["desc"][i], np.uint8), tdesc[j])
            if d < best:
                best, bidx = d, j
        bd[i] = best
        if best <= 50:
            bi[i] = bidx
    return bi, bd


def test_fuse_search_equals_the_restatement():
    from golden_cases import fuse_cases
    cases = [pr for _, pr in fuse_cases()]
    for seed, kw in ((81, {"n": 400, "m": 300, "th": 5.0}), (82, {"n": 2500, "m": 1200})):
        pr = synth.fuse_scene(seed, **kw)
        from pointslot_amd.matcher import build_grid
        pr["train"]["cell_off"], pr["train"]["cell_idx"] = build_grid(pr["train"]["x"], pr["train"]["y"], *pr["train"]["grid"])   # (the restatement's input format)
        cases.append(pr)
    for k, pr in enumerate(cases):
        bi, bd = _fuse(pr)
        oi, od = oracle_lib.fuse_search(pr)
        assert np.array_equal(bi, oi) and np.array_equal(bd, od), (k, int((bi != oi).sum()), int((bd != od).sum()))
        assert (bi >= 0).sum() > 10
        # the other reading of `Rco * Poj + tco` (float accumulators): no candidate's match changes on these scenes
        bi2, bd2 = _fuse(pr, accumulate="float")
        assert np.array_equal(bi2, bi) and np.array_equal(bd2, bd), k
