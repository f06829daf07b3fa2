"""Developer tool: lockstep tracking throughput of examples/stereo_kitti_batch for several batch sizes.
usage: [PS_GROUPS=G] python tools/batch_track_bench.py [frames] [S ...]"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd import sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "stereo_kitti_batch")
GROUPS = os.environ.get("PS_GROUPS", "1")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sizes = [int(a) for a in sys.argv[2:]] or [8, 32]
tmp = tempfile.mkdtemp(prefix="psbatch_")
dirs = []
for k in range(8):
    seq = sequence.generate(n_frames=frames, seed=30 + k, step=0.05 + 0.01 * k)
    d = os.path.join(tmp, "%04d" % k)
    sequence.write_pgm(d, seq)
    dirs.append(d)
for S in sizes:
    out = subprocess.run([EXE, "--groups", GROUPS] + [dirs[i % len(dirs)] for i in range(S)], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(out.stdout[-2000:], out.stderr[-2000:])
        sys.exit(1)
    if os.environ.get("PS_VERBOSE"):
        print(out.stdout)
    print(json.loads(out.stdout.strip().splitlines()[-1]))
