"""Developer tool: per-frame tracking time of the two C++ drivers on one generated sequence."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointslot_amd import sequence
d = tempfile.mkdtemp()
seq = sequence.generate(n_frames=12, seed=30, step=0.05)
sequence.write_pgm(d, seq)
for exe in ("build/stereo_kitti", "build/stereo_kitti_batch"):
    out = subprocess.run([os.path.join(ROOT, exe), d], capture_output=True, text=True)
    print(exe, [l for l in out.stdout.splitlines() if "median" in l or "mean" in l][-2:])
