"""The HIP path (through the C-ABI) against the committed matcher / optimiser fixtures: the same checks the CPU checker has
to pass in tests/test_golden_cpu.py, without the checker in the loop."""
import json
import os

import numpy as np
import pytest

from golden_cases import matcher_cases, pose_cases, ba_cases, cfse3_cases, fuse_cases, distinctive_case, dynamic_cases, stereo_case
from test_golden_cpu import check_match, check_pose, check_ba, check_aux, GOLD

pytestmark = pytest.mark.gpu


def test_matcher_fixtures_on_the_gpu():
    from pointslot_amd.matcher import ORBmatcher
    gold = json.load(open(os.path.join(GOLD, "match_golden.json")))
    for name, kind, pr, arg in matcher_cases():
        if kind == "bruteforce":
            m = ORBmatcher(arg[0], arg[1])
            (n, out), = m.SearchByBruceMatching([pr])
        elif kind == "frame":
            m = ORBmatcher(0.9, arg)
            (n, out), = m.SearchByProjection([pr])
        else:
            m = ORBmatcher(arg, True)
            (n, out), = m.SearchByProjection([pr])
        check_match(gold[name], n, np.asarray(out))
        m.close()


def test_optimiser_fixtures_on_the_gpu():
    from pointslot_amd.optimizer import Optimizer
    gold = json.load(open(os.path.join(GOLD, "opt_golden.json")))
    opt = Optimizer()
    opt.enable_trace(True)
    cases = pose_cases()
    res = opt.PoseOptimization([p for _, p in cases])
    for k, (name, _) in enumerate(cases):
        r, tcw, outlier = res[k]
        check_pose(gold[name], r, tcw, np.asarray(outlier), opt.get_trace(k), strict=False)
    for name, p in ba_cases():
        r, = opt.ObjectLocalBundleAdjustment([p])
        check_ba(gold[name], r["n_erased"], r["poses"], r["points"], r["erase"], r["trace"], strict=False)
    opt.close()


def test_further_fixtures_on_the_gpu():
    from pointslot_amd import optimizer as optmod
    from pointslot_amd.extractor import ORBextractor, ComputeStereoMatches
    from pointslot_amd.matcher import ORBmatcher
    from pointslot_amd.optimizer import Optimizer
    gold = json.load(open(os.path.join(GOLD, "aux_golden.json")))
    opt = Optimizer()
    cases = cfse3_cases(optmod.se3_from_mat4f)
    res = opt.CFSE3ObjStateOptimization([f for _, f in cases])
    cf = {name: res[k] for k, (name, _) in enumerate(cases)}
    dyn = opt.DynamicStaticDiscrimination(dynamic_cases())
    opt.close()
    m = ORBmatcher(0.6, True)
    fcases = fuse_cases()
    fres = m.FuseSearch([pr for _, pr in fcases])
    fu = {name: fres[k] for k, (name, _) in enumerate(fcases)}
    dist = m.ComputeDistinctiveDescriptors(distinctive_case())
    m.close()
    L, R = stereo_case()
    exl, exr = ORBextractor(2000, 1.2, 8, 20, 5), ORBextractor(2000, 1.2, 8, 20, 5)
    exl(L); exr(R)
    ur, dp = ComputeStereoMatches(exl, exr, np.float32(384.38148 / 721.5377), np.float32(384.38148))
    exl.close(); exr.close()
    check_aux(gold, cf, fu, dist, dyn, (int((ur >= 0).sum()), ur, dp), strict=False)
