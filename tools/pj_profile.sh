python3 - <<PY
import sys
sys.path.insert(0, ".")
from pointslot_amd import sequence
sequence.write_pgm("/tmp/ps_one", sequence.generate(n_frames=6, seed=30, step=0.05))
PY
LD_PRELOAD=$PWD/build_exp/libps_pjprof.so build/stereo_kitti /tmp/ps_one | grep -a pj_resolve | tail -6
