#!/bin/bash
# The drop-in path as a maintainer's build would run it: examples/stereo_kitti.cpp (ORBextractor x 2 threads, ComputeStereoMatches,
# SearchByProjection, PoseOptimization through the shim classes, host images) over a generated sequence on disk.
#   tools/shim_path_bench.sh [frames]
N=${1:-40}
D=/tmp/ps_shim_seq
python3 - <<PY
import sys
sys.path.insert(0, "$PWD")
from pointslot_amd import sequence
q = sequence.generate_drive(n_frames=$N, seed=40, texture=sequence.kitti_texture())
sequence.write_pgm("$D", q)
PY
./build/stereo_kitti $D | tail -6
