"""Randomised parity sweeps as part of the GPU suite: the developer tools under tools/stress_*.py - random image sizes, strides, pyramid
parameters, masks, matcher scenes and optimiser graphs against the CPU checker, bit for bit - run here with a fixed seed and a small
budget each (a few seconds), so that every GPU test run also covers shapes no hand-written case has (tile / cell / tap edge cases of the
extraction kernels in particular).  The tools exit non-zero on the first difference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEPS = [
    ("stress_orb.py", ["20250501", "60"], "0 mismatches"),                    # ORBextractor: 60 random images / parameter sets
    ("stress_cvorb_batch.py", ["20250502", "12"], "identical to the CPU restatement"),   # batched cv::ORB stand-in under random masks
    ("stress_matchers.py", ["20250503"], "0 mismatches"),                     # the matchers' scenes
    ("stress_opt.py", ["20250504"], "0 mismatches"),                          # PoseOptimization / CFSE3 / object BA graphs
]


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args,expect", SWEEPS, ids=[s[0][:-3] for s in SWEEPS])
def test_random_sweep(tool, args, expect):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert expect in r.stdout, tail
