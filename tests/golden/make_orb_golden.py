#!/usr/bin/env python3
"""Generates tests/golden/orb_golden.json: per-stage digests of the CPU oracle (oracle/orb_oracle.cpp)
on the committed inputs.  The reference cannot be built here (OpenCV absent), so these vectors pin the
oracle against regressions — they are NOT outputs of the reference (parity unpinned)."""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_lib import OracleORB  # noqa: E402
from pointslot_amd import synth  # noqa: E402

imgs = {"synth_left": synth.stereo_pair()[0],
        "kitti_000212": np.array(Image.open(os.path.join(HERE, "kitti_000212_gray.png")))}
out = {}
for image in imgs:
    for nf in (2000, 1000):
        o = OracleORB(nf)
        kps, desc = o.run(imgs[image])
        out["%s_%d" % (image, nf)] = {
            "image": image, "nfeatures": nf, "n": int(len(kps)),
            "kps_sha256": hashlib.sha256(kps.tobytes()).hexdigest(),
            "desc_sha256": hashlib.sha256(desc.tobytes()).hexdigest(),
            "ncand": [int(len(o.candidates(l))) for l in range(8)],
            "nsel": [int(len(o.level_keypoints(l))) for l in range(8)],
            "pyr_sha": [hashlib.sha256(o.padded(l).tobytes()).hexdigest()[:16] for l in range(8)],
            "blur_sha": [hashlib.sha256(o.blur(l).tobytes()).hexdigest()[:16] for l in range(8)],
        }
json.dump(out, open(os.path.join(HERE, "orb_golden.json"), "w"), indent=1)
print("wrote", len(out), "entries")
