"""examples/stereo_kitti.cpp — the reference's stereo driver shape on the C++ host shim (ORBextractor / ORBmatcher / Optimizer
classes + StereoOdometry.h): compiles with plain g++ against the C-ABI, and on a GPU box tracks a generated sequence in the
reference's on-disk layout to the same trajectory as the Python driver of the same kernels."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "stereo_kitti")
EXE_BATCH = os.path.join(ROOT, "tests", "cpp", "stereo_kitti_batch")


def _build(exe=EXE):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "pointslot_amd", "host"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", os.path.basename(exe) + ".cpp"), "-o", exe, "-L", os.path.join(ROOT, "pointslot_amd"),
           "-lpointslot_hip", "-pthread", "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"), "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)


@pytest.mark.parametrize("exe", [EXE, EXE_BATCH])
def test_stereo_kitti_cpp_compiles_and_links(exe):
    _build(exe)
    assert os.path.exists(exe)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 1 and "Usage" in out.stderr


def test_cpp_host_state_machine_with_the_cpu_checker_behind_it(tmp_path):
    """OdoSequence (the C++ host side of the tracking loop: request preparation, match application, outlier handling, motion
    model) with oracle/liboracle.so serving its SearchByProjection / PoseOptimization requests, against the Python twin of the
    loop over the same checker: same match counts, same trajectory.  No GPU involved."""
    import oracle_lib  # noqa: F401  (builds oracle/liboracle.so when missing)
    from pointslot_amd import sequence
    from pointslot_amd.tracker import StereoOdometry
    from oracle_backend import OracleBackend
    exe = os.path.join(ROOT, "tests", "cpp", "odo_oracle_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "pointslot_amd", "host"), "-I", os.path.join(ROOT, "include"),
                           exe + ".cpp", "-o", exe, "-L", os.path.join(ROOT, "oracle"), "-loracle", "-L", os.path.join(ROOT, "pointslot_amd"), "-lpointslot_hip",
                           "-pthread", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    seq = sequence.generate(n_frames=5, seed=4, w=800, h=300)
    d = str(tmp_path / "0000")
    sequence.write_pgm(d, seq)
    out = subprocess.run([exe, d], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-1000:]
    traj = np.loadtxt(os.path.join(d, "CameraTrajectoryOracle.txt"))
    assert traj.shape == (5, 12)
    vo = StereoOdometry(OracleBackend(), seq["K"], seq["bf"], 800, 300)
    for l, r in zip(seq["left"], seq["right"]):
        vo.track(l, r)
    py = np.array([np.concatenate([t[:3, :3].T, (-(t[:3, :3].T @ t[:3, 3]))[:, None]], 1).reshape(12) for t in vo.trajectory])
    assert np.abs(py - traj).max() < 1e-6, np.abs(py - traj).max()      # host float arithmetic may differ in the last ulp
    assert np.abs(traj[:, 3] - seq["twc"][:, 0, 3]).max() < 0.03
    lines = [l for l in out.stdout.splitlines() if l.startswith("frame")]
    for k in range(1, 5):
        st = vo.stats[k]
        assert "matches %d map %d local inliers %d" % (st["matches"], st["map_matches"], st["local_inliers"]) in lines[k], (lines[k], st)


def test_cpp_host_state_machine_under_address_and_ub_sanitizers(tmp_path):
    """The same CPU-only driver built with -fsanitize=address,undefined (sanitizers run on the CPU build only): the host side of
    the tracking loop indexes a dozen parallel arrays per frame; no report, same exit code."""
    import oracle_lib  # noqa: F401
    from pointslot_amd import sequence
    exe = str(tmp_path / "odo_oracle_driver_asan")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "pointslot_amd", "host"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "odo_oracle_driver.cpp"), "-o", exe, "-L", os.path.join(ROOT, "oracle"),
                        "-loracle", "-L", os.path.join(ROOT, "pointslot_amd"), "-lpointslot_hip", "-pthread", "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                        "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"), "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    seq = sequence.generate(n_frames=4, seed=5, w=800, h=300)
    d = str(tmp_path / "0000")
    sequence.write_pgm(d, seq)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, d], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]
    assert out.stdout.count(": ok ") == 4


@pytest.mark.gpu
def test_stereo_kitti_cpp_tracks_like_the_python_driver(tmp_path):
    from pointslot_amd import sequence
    from pointslot_amd.tracker import HipBackend, StereoOdometry
    if not os.path.exists(EXE):
        _build()
    seq = sequence.generate(n_frames=8, seed=4)
    d = str(tmp_path / "0000")
    sequence.write(d, seq, pgm=True)
    out = subprocess.run([EXE, d], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "trajectory saved!" in out.stdout and "LOST" not in out.stdout
    traj = np.loadtxt(os.path.join(d, "CameraTrajectory.txt"))
    assert traj.shape == (8, 12)
    # ground truth of the generator
    assert np.abs(traj[:, 3] - seq["twc"][:, 0, 3]).max() < 0.03 and np.abs(traj[:, [7, 11]]).max() < 0.03
    # the Python driver over the same kernels (host float arithmetic differs in the last ulp: tolerance, not bit-equality)
    be = HipBackend()
    h, w = seq["left"][0].shape
    vo = StereoOdometry(be, seq["K"], seq["bf"], w, h)
    for l, r in zip(seq["left"], seq["right"]):
        vo.track(l, r)
    be.close()
    py = np.array([np.concatenate([t[:3, :3].T, (-(t[:3, :3].T @ t[:3, 3]))[:, None]], 1).reshape(12) for t in vo.trajectory])
    assert np.abs(py - traj).max() < 2e-3, np.abs(py - traj).max()


@pytest.mark.gpu
def test_lockstep_batch_tracks_every_sequence_like_the_single_sequence_driver(tmp_path):
    """StereoOdometryBatch (one batched extraction per step, one C-ABI call per round of search / pose problems) against
    StereoOdometry (the reference's per-frame call structure) on the same sequences: identical trajectory files.  One of the
    sequences is blank (no keypoints: it never initialises) and must not disturb the others."""
    import json
    from pointslot_amd import sequence
    _build(EXE)
    _build(EXE_BATCH)
    dirs = []
    for k, seed in enumerate((4, 9, 21)):
        seq = sequence.generate(n_frames=6, seed=seed, step=0.06 + 0.02 * k)
        d = str(tmp_path / ("%04d" % k))
        sequence.write(d, seq, pgm=True)
        dirs.append(d)
    blank = dict(seq)
    blank["left"] = np.full_like(seq["left"], 128); blank["right"] = np.full_like(seq["right"], 128)
    dblank = str(tmp_path / "blank")
    sequence.write(dblank, blank, pgm=True)
    for d in dirs:
        out = subprocess.run([EXE, d], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    order = [dirs[0], dblank, dirs[1], dirs[2]]
    out = subprocess.run([EXE_BATCH] + order, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["sequences"] == 4 and stats["frames_per_sequence"] == 6 and stats["untracked_frames"] == 6   # the blank one
    for d in dirs:
        single = open(os.path.join(d, "CameraTrajectory.txt")).read()
        batch = open(os.path.join(d, "CameraTrajectoryBatch.txt")).read()
        assert len(single.splitlines()) == 6
        assert single == batch, d
    assert open(os.path.join(dblank, "CameraTrajectoryBatch.txt")).read() == ""
