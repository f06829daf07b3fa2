#!/usr/bin/env python3
"""Generates tests/golden/match_golden.json and opt_golden.json: outputs of the CPU oracle (oracle/match_oracle.cpp,
oracle/opt_oracle.cpp) on the seeded synthetic problems of pointslot_amd.synth (SURVEY.md 8c "golden vectors to commit",
items iii and iv).  The reference cannot be built here (OpenCV / Eigen absent), so these vectors pin the oracle against
regressions and give the GPU path a fixed target - they are NOT outputs of the reference (parity unpinned).
Index / mask / count results are exact; FP64 results are stored as numbers and compared with a tolerance."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib  # noqa: E402
from golden_cases import matcher_cases, pose_cases, ba_cases, cfse3_cases, fuse_cases, distinctive_case, dynamic_cases, stereo_case  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


match = {}
for name, kind, pr, arg in matcher_cases():
    if kind == "bruteforce":
        n, out = oracle_lib.search_bruteforce(pr, *arg)
    elif kind == "frame":
        n, out = oracle_lib.search_projection_frame(pr, check_ori=arg)
    else:
        n, out = oracle_lib.search_projection_points(pr, arg)
    match[name] = {"nmatches": int(n), "assigned": int((out >= 0).sum()), "match_sha256": sha(out.astype(np.int32)),
                   "first_assignments": [[int(j), int(out[j])] for j in np.nonzero(out >= 0)[0][:12]]}
json.dump(match, open(os.path.join(HERE, "match_golden.json"), "w"), indent=1)

opt = {}
for name, p in pose_cases():
    r, tcw, outlier, trace = oracle_lib.pose_optimize(p, want_trace=True)
    opt[name] = {"result": int(r), "tcw": [float(v) for v in tcw.reshape(16)], "outlier_sha256": sha(outlier.astype(np.uint8)),
                 "n_outliers": int(outlier.sum()), "trace": [[float(c), float(l), int(t)] for c, l, t in trace]}
for name, p in ba_cases():
    n, poses, pts, erase, trace = oracle_lib.object_ba(p)
    opt[name] = {"erased": int(n), "erase_sha256": sha(erase.astype(np.uint8)), "poses": [[float(v) for v in row] for row in poses],
                 "points_sum": [float(v) for v in pts.sum(0)], "points_first": [[float(v) for v in row] for row in pts[:4]],
                 "trace": [[float(c), float(l), int(t)] for c, l, t in trace]}
json.dump(opt, open(os.path.join(HERE, "opt_golden.json"), "w"), indent=1)
aux = {}
for name, f in cfse3_cases(oracle_lib.se3_from_mat4f):
    ok, poses, outl = oracle_lib.cfse3_optimize(f["objs"], f["K"])
    aux[name] = {"ok": int(ok), "poses": [[float(v) for v in row] for row in poses], "outlier_sha256": [sha(o.astype(np.uint8)) for o in outl],
                 "n_outliers": [int(o.sum()) for o in outl]}
for name, pr in fuse_cases():
    bi, bd = oracle_lib.fuse_search(pr)
    aux[name] = {"best_idx_sha256": sha(np.asarray(bi, np.int32)), "best_dist_sha256": sha(np.asarray(bd, np.int32)), "n_found": int((np.asarray(bi) >= 0).sum())}
aux["distinctive"] = {"best": [int(v) for v in oracle_lib.distinctive_descriptors(distinctive_case())]}
aux["dynamic"] = [[float(r[0]).hex(), float(r[1]).hex(), int(r[2]), int(r[3])] for r in (oracle_lib.dynamic_discrimination(o) for o in dynamic_cases())]
L, R = stereo_case()
ol, orr = oracle_lib.OracleORB(2000), oracle_lib.OracleORB(2000)
ol.run(L); orr.run(R)
kept, ur, dp = oracle_lib.stereo_match(ol, orr, np.float32(384.38148 / 721.5377), np.float32(384.38148))
aux["stereo"] = {"kept": int(kept), "n": int(len(ur)), "u_right_sha256": sha(ur.astype(np.float32)), "depth_sha256": sha(dp.astype(np.float32))}
json.dump(aux, open(os.path.join(HERE, "aux_golden.json"), "w"), indent=1)
print("wrote", len(match), "matcher,", len(opt), "optimiser and", len(aux), "further entries")
