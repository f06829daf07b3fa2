#!/usr/bin/env python3
"""Usage: stereo_kitti.py path_to_sequence [max_frames]      (cf. /root/reference/Examples/Stereo/stereo_kitti.cc:55-166)

Runs the GPU hot path over a stereo sequence in the reference's on-disk layout (image_02 / image_03 / timestamp.txt) and
writes CameraTrajectory.txt in the System::SaveTrajectoryKITTI format.  `--generate N` first writes the generated
mini-sequence of pointslot_amd.sequence into the directory."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from pointslot_amd import sequence                                    # noqa: E402
from pointslot_amd.tracker import HipBackend, StereoOdometry, save_trajectory_kitti   # noqa: E402
from pointslot_amd.synth import KITTI_K, KITTI_BF                      # noqa: E402


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 1
    seq_dir = argv[1]
    if "--generate" in argv:
        n = int(argv[argv.index("--generate") + 1])
        sequence.write(seq_dir, sequence.generate(n_frames=n))
    nmax = int(argv[2]) if len(argv) > 2 and argv[2].isdigit() else None
    seq = sequence.load(seq_dir, nmax)
    c = seq["calib"]
    K = (c.get("Camera.fx", KITTI_K[0]), c.get("Camera.fy", KITTI_K[1]), c.get("Camera.cx", KITTI_K[2]), c.get("Camera.cy", KITTI_K[3]))
    bf = c.get("Camera.bf", KITTI_BF)
    h, w = seq["left"][0].shape
    vo = StereoOdometry(HipBackend(), K, bf, w, h, th_depth=c.get("ThDepth", 35.0))
    times = []
    for k, (l, r) in enumerate(zip(seq["left"], seq["right"])):
        t0 = time.perf_counter()
        tcw = vo.track(l, r)
        times.append(time.perf_counter() - t0)
        print("frame %d: %s %s" % (k, "LOST" if tcw is None else "ok", vo.stats[-1]))
    times.sort()
    print("median tracking time: %.3f ms, mean %.3f ms" % (1e3 * times[len(times) // 2], 1e3 * sum(times) / len(times)))
    save_trajectory_kitti(os.path.join(seq_dir, "CameraTrajectory.txt"), vo.trajectory)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
