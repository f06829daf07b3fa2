#!/usr/bin/env python3
"""Damping trials per LM iteration (g2o's qmax, levenberg.cpp:102-149) in BASELINE config 3 (64 frames x 2000 stereo edges, 4 x 10
schedule) and config 4 (8 objects x 50 KF x 300 points, 5 + 10), from the CPU checker's per-iteration traces - the same traces the
GPU tests compare trial by trial.  Runs on the CPU (no GPU): python tools/trial_histogram.py > profiles/r05_trial_histogram.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402
from pointslot_amd import synth  # noqa: E402


def hist(traces, title):
    h = collections.Counter()
    pos = collections.Counter()           # (position of the iteration from the END of its optimize() call, trials)
    n_it = n_tr = 0
    for t in traces:
        q = t[:, 2].astype(int)
        n_it += len(q); n_tr += int(q.sum())
        for v in q:
            h[int(v)] += 1
    print("%s: %d LM iterations, %d damping trials" % (title, n_it, n_tr))
    print("  trials per iteration : " + "  ".join("%d:%d" % (k, h[k]) for k in sorted(h)))
    rej = sum((k - 1) * v for k, v in h.items() if k < 10) + 10 * h.get(10, 0)
    print("  rejected trials      : %d (%.0f %% of all trials); in iterations that end after 10 rejections: %d" % (rej, 100.0 * rej / max(n_tr, 1), 10 * h.get(10, 0)))


def main():
    tr = []
    for k in range(64):
        p = synth.pose_problem(0x51070003 + k)
        tr.append(oracle_lib.pose_optimize(p, want_trace=True)[3])
    hist(tr, "config 3 (PoseOptimization, 64 frames x 2000 stereo edges)")
    # where in a call: the trace of a call is its 4 rounds back to back; rounds end at an iteration with 10 trials, a zero gain or 3 bad iterations
    tr = []
    for j in range(8):
        g = synth.object_ba_problem(0x51070004 + j)
        tr.append(oracle_lib.object_ba(g)[4])
    hist(tr, "config 4 (ObjectLocalBundleAdjustment, 8 objects x 50 KF x 300 points, p = 1.0)")
    for t in tr[:2]:
        print("  object trace (chi2, lambda, trials): " + " | ".join("%.6g %.3g %d" % (a, b, c) for a, b, c in t))


if __name__ == "__main__":
    main()
