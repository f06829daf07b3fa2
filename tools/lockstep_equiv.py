"""Developer tool: many generated sequences through the single-sequence driver and through the lockstep driver (several groups);
the trajectory files must be identical.  usage: python tools/lockstep_equiv.py [n_sequences] [frames] [groups]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointslot_amd import sequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
groups = sys.argv[3] if len(sys.argv) > 3 else "3"
tmp = tempfile.mkdtemp(prefix="ps_equiv_")
dirs = []
for k in range(n):
    seq = sequence.generate(n_frames=frames, seed=500 + k, step=0.04 + 0.015 * (k % 5), n_boxes=k % 3)
    d = os.path.join(tmp, "%04d" % k)
    sequence.write_pgm(d, seq)
    dirs.append(d)
for d in dirs:
    subprocess.run([os.path.join(ROOT, "build", "stereo_kitti"), d], capture_output=True, check=True)
out = subprocess.run([os.path.join(ROOT, "build", "stereo_kitti_batch"), "--groups", groups] + dirs, capture_output=True, text=True)
assert out.returncode == 0, out.stderr[-500:]
bad = 0
for d in dirs:
    a = open(os.path.join(d, "CameraTrajectory.txt")).read(); b = open(os.path.join(d, "CameraTrajectoryBatch.txt")).read()
    if a != b or len(a.splitlines()) != frames:
        bad += 1
        print("DIFFERENT", d, len(a.splitlines()), len(b.splitlines()))
print("lockstep equivalence: %d sequences x %d frames in %s groups, %d differing" % (n, frames, groups, bad))
