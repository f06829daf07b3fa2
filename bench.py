#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on N MI355X of one node.

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  the ORB extractor (8-level pyramid + 7x7 blur fused per level, FAST cells, quadtree, rBRIEF; BASELINE.json configs[1])
  over a batch of PAIRS stereo pairs 1242x375, 2000 keypoints per image.
`value` = stereo frames/s over all ranks (weak scaling: every rank owns its own batch; the path shards
by image, so there is no data-path collective).  Rank 0 prints ONE JSON line.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMG_W, IMG_H, NFEAT = 1242, 375, 2000
# SURVEY.md section 8d / BASELINE.md section 3: algorithmic bytes per image, unfused, each buffer touched once
LEVEL_PX = 1441432            # sum of the 8 level areas for 1242x375
PADDED_PX = 1735932           # same with the 19-px borders
ALGO_BYTES_PER_IMAGE = {
    # the fused level kernel (pyramid + border + blur in one launch per level) is priced with the SUM of the two unfused
    # stages it replaces, as SURVEY.md defines the per-image figure (its own traffic is lower: the blur input stays in LDS)
    "orb_level_fused": 465750 + PADDED_PX + PADDED_PX + LEVEL_PX,
    "orb_pyramid_level": 465750 + PADDED_PX,          # input read + padded pyramid write
    "orb_fast_cells": PADDED_PX,                      # FAST reads the padded pyramid once
    "orb_quadtree": 0,                                # candidate lists only (not in the pixel budget)
    "orb_blur": PADDED_PX + LEVEL_PX,                 # blur read + blur write
    "orb_describe": LEVEL_PX + NFEAT * (32 + 28),     # gather (upper bound) + outputs
}
RED_DEV = "cuda"            # device of the tensors used for cross-rank reductions
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def pmc_traffic(kernel, nimg):
    """HBM bytes per launch of `kernel` from the committed PMC measurement (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes, same workload).  On gfx950 FETCH_SIZE tallies 64 B per 128-B request of a coalesced
    stream, so it is doubled as MI355X_MICROARCH.md prescribes; WRITE_SIZE is taken as reported (uncalibrated).  Returns None
    when no measurement for this launch shape is on file."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None
    if t.get("images_per_launch") != nimg:
        return None
    for name, v in t.get("kernels", {}).items():
        if name.split("<")[0] == kernel:
            return v.get("hbm_bytes_per_launch_fetch_doubled")
    return None


def pmc_valu_issue(kernel, nimg):
    """Share of the launch that VALU issue alone accounts for (waves x VALU instructions x 4 cycles / 1024 SIMDs / clock), from the
    committed PMC pass (tools/pmc_issue.sh: SQ_INSTS_VALU, SQ_WAVES, own run).  Explains a low HBM fraction: the kernel is bound by
    instruction issue, not by memory.  None when no measurement for this launch shape is on file."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_valu_issue.json")))
    except (OSError, ValueError):
        return None
    if t.get("images_per_launch") != nimg:
        return None
    for name, v in t.get("kernels", {}).items():
        if name.split("<")[0] == kernel:
            return {"valu_issue_share": v.get("valu_issue_share"), "valu_instructions_per_wave": v.get("valu_per_wave"), "waves_per_launch": v.get("waves_per_launch")}
    return None


def cpu_baseline(batch, pairs_sample):
    """Times the CPU restatement (oracle/, kind "port") on a bounded sample of the same workload, using the
    reference's thread model: left and right image on two threads (/root/reference/src/Frame.cc:709-710)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    from oracle_lib import OracleORB
    orcs = [OracleORB(NFEAT), OracleORB(NFEAT)]
    n = min(pairs_sample, batch.shape[0] // 2)

    def run(side):
        for k in range(n):
            orcs[side].run(batch[2 * k + side])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(2) as ex:
        list(ex.map(run, [0, 1]))
    dt = time.perf_counter() - t0
    # SURVEY.md 8d also asks for the "all host cores" rate: independent images over every core, one extractor object per thread
    ncore = min(os.cpu_count() or 1, batch.shape[0])
    pool = [OracleORB(NFEAT) for _ in range(ncore)]
    reps = 2

    def run_all(c):
        for r in range(reps):
            pool[c].run(batch[(c + r * ncore) % batch.shape[0]])

    t1 = time.perf_counter()
    with ThreadPoolExecutor(ncore) as ex:
        list(ex.map(run_all, range(ncore)))
    dt_all = time.perf_counter() - t1
    return {"value": n / dt, "unit": "frames/s", "cores": 2, "kind": "port",
            "sample": "%d stereo pairs of the step's batch, CPU restatement of the reference algorithm "
                      "(oracle/orb_oracle.cpp, -O3 -march=native), left/right on 2 threads; host has %d cores"
                      % (n, os.cpu_count()),
            "all_cores": {"value": ncore * reps / 2 / dt_all, "unit": "frames/s", "cores": ncore,
                          "sample": "%d images, independent images on %d threads" % (ncore * reps, ncore)}}


def secondary_metrics(rank, world, local_rank, dist, with_cpu):
    """BASELINE configs[2] and [3]: per-frame pose optimisation (batch of 64 frames x 2000 stereo edges) and the
    object local BA (8 objects x 50 keyframes x 300 points).  Frames / objects are independent units and are
    sharded over the ranks; times are max-over-ranks."""
    from pointslot_amd import parallel, synth
    from pointslot_amd.optimizer import Optimizer
    opt = Optimizer(device=local_rank)
    out = {}
    # ---- pose optimisation: 64 frames total ----
    mine = list(parallel.shard_units(64, world, rank))
    frames = [synth.pose_problem(0x51070003 + k) for k in mine]
    opt.PoseOptimization(frames[:1])                       # warm-up
    t0 = time.perf_counter()
    res = opt.PoseOptimization(frames)
    wall = time.perf_counter() - t0
    kern_ms = opt.last_kernel_ms()
    wall = parallel.max_over_ranks(dist, wall, RED_DEV)
    kern_ms = parallel.max_over_ranks(dist, kern_ms, RED_DEV)
    out["pose_optimization"] = {"workload": "BASELINE configs[2]: 64 frames x (1 SE3 x 2000 stereo edges), 4 x 10 LM schedule",
                                "frames": 64, "kernel_ms_per_batch": kern_ms, "wall_ms_per_batch_incl_pcie": wall * 1e3,
                                "frames_per_s_kernel": 64 / (kern_ms * 1e-3), "inliers_frame0": int(res[0][0]) if res else None}
    # ---- object BA: 8 objects total ----
    mine = list(parallel.shard_units(8, world, rank))
    graphs = [synth.object_ba_problem(0x51070004 + j, perturb=(0.05, 1.0, 0.02), perturb_axis="z") for j in mine]
    if graphs:
        opt.ObjectLocalBundleAdjustment(graphs[:1])        # warm-up (allocations)
        r = opt.ObjectLocalBundleAdjustment(graphs)
        ms = opt.last_kernel_ms()
        iters = max(x["iterations"] for x in r)
        trials = max(x["trials"] for x in r)
        r1 = opt.ObjectLocalBundleAdjustment(graphs[:1])    # SURVEY.md 8d config 4 also asks for one object alone
        ms1, iters1 = opt.last_kernel_ms(), max(r1[0]["iterations"], 1)
    else:
        ms, iters, trials = 0.0, 1, 1
        ms1, iters1 = 0.0, 1
    ms = parallel.max_over_ranks(dist, ms, RED_DEV)
    iters = int(parallel.max_over_ranks(dist, iters, RED_DEV))
    out["object_ba"] = {"workload": "BASELINE configs[3]: 8 objects x 50 ObjectKeyFrames x 300 MapObjectPoints (15 000 stereo edges each), "
                                    "Schur LM 5 + 10 iterations", "objects": 8, "gpu_ms_per_batch": ms, "lm_iterations": iters,
                        "lm_trials": int(parallel.max_over_ranks(dist, trials, RED_DEV)), "ms_per_iter": ms / max(iters, 1),
                        "ms_per_iter_1_object": parallel.max_over_ranks(dist, ms1 / iters1, RED_DEV)}
    if with_cpu and rank == 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        t0 = time.perf_counter()
        for f in frames[:16]:
            oracle_lib.pose_optimize(f)
        out["pose_optimization"]["cpu_port_frames_per_s_1core"] = 16 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        n, _, _, _, tr = oracle_lib.object_ba(graphs[0])
        dt = time.perf_counter() - t0
        out["object_ba"]["cpu_port_ms_per_iter_1core_1object"] = dt * 1e3 / max(len(tr), 1)
    opt.close()
    out["sequence_tracking"] = sequence_leg(rank, world, local_rank, dist, with_cpu)
    out["lockstep_tracking"] = lockstep_leg(rank, world, local_rank, dist)
    if rank == 0:
        try:
            out["next_rows"] = next_rows_leg(local_rank)
        except Exception as e:   # noqa: BLE001  (a failure of this leg must not take the bench line down)
            out["next_rows"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def next_rows_leg(local_rank):
    """SURVEY.md 8f rows 3 and 4, one wall-clock figure each on rank 0 (host buffers, PCIe included; the parity of every
    row is in tests/): LocalBundleAdjustment, ComputeDistinctiveDescriptors, the search half of Fuse, and the
    DynamicStaticDiscrimination reprojection test."""
    from pointslot_amd import synth
    from pointslot_amd.matcher import ORBmatcher, build_grid
    from pointslot_amd.optimizer import Optimizer
    out = {}
    opt = Optimizer(device=local_rank)
    # f-3: a local-BA shaped graph - 10 free + 6 fixed keyframes (VertexSE3Expmap), 1200 world points, 35 % visibility
    g = synth.object_ba_problem(0x51070060, n_kf=10, n_pts=1200, p_vis=0.35, perturb=(0.05, 1.0, 0.03), perturb_axis="y", n_fixed_extra=6, mono_frac=0.15)
    g["pose_flags"] = (g["pose_flags"] & 1).astype(np.uint8)
    opt.ObjectLocalBundleAdjustment([g])
    r, = opt.ObjectLocalBundleAdjustment([g])
    out["local_ba"] = {"workload": "f-3: 16 keyframes (6 fixed) x 1200 points, %d edges" % len(g["e_pose"]), "gpu_ms": opt.last_kernel_ms(),
                       "lm_iterations": int(r["iterations"]), "ms_per_iter": opt.last_kernel_ms() / max(int(r["iterations"]), 1)}
    # f-4: DynamicStaticDiscrimination on 8 detections x 300 object points
    objs = [synth.dynamic_object(100 + k, n=300, moving=0.1 * k) for k in range(8)]
    opt.DynamicStaticDiscrimination(objs)
    t0 = time.perf_counter()
    for _ in range(10):
        opt.DynamicStaticDiscrimination(objs)
    out["dynamic_static_discrimination"] = {"workload": "f-4: 8 detections x 300 object points per call", "wall_ms_per_call": (time.perf_counter() - t0) * 100}
    opt.close()
    m = ORBmatcher(0.6, True, device=local_rank)
    # f-4: ComputeDistinctiveDescriptors of 2400 map points with 2..50 observations each
    rng = np.random.default_rng(7)
    lists = [rng.integers(0, 256, (int(n), 32), dtype=np.uint8) for n in rng.integers(2, 51, 2400)]
    m.ComputeDistinctiveDescriptors(lists)
    t0 = time.perf_counter()
    m.ComputeDistinctiveDescriptors(lists)
    dt = time.perf_counter() - t0
    out["distinctive_descriptors"] = {"workload": "f-4: 2400 map points, 2..50 observations each (%d descriptors)" % sum(len(x) for x in lists),
                                      "wall_ms_per_call": dt * 1e3, "points_per_s": len(lists) / dt}
    # f-4: the search half of Fuse, 6 keyframes x (1500 features, 800 candidate points)
    prs = []
    for k in range(6):
        pr = synth.fuse_scene(60 + k, n=1500, m=800)
        T = pr["train"]
        T["cell_off"], T["cell_idx"] = build_grid(T["x"], T["y"], *T["grid"])
        prs.append(pr)
    m.FuseSearch(prs)
    t0 = time.perf_counter()
    m.FuseSearch(prs)
    out["fuse_search"] = {"workload": "f-4: 6 keyframes x (1500 features, 800 candidate points) per call", "wall_ms_per_call": (time.perf_counter() - t0) * 1e3}
    m.close()
    return out


def lockstep_leg(rank, world, local_rank, dist, n_sequences=128, n_groups=2, n_frames=12, n_distinct=4, n_procs=2):
    """BASELINE config 4 at batch scale: every rank tracks `n_sequences` independent stereo sequences in lockstep with the
    C++ host driver (examples/stereo_kitti_batch.cpp -> StereoOdometryBatch: one batched extraction + stereo matching per
    step, one C-ABI call per round of SearchByProjection / PoseOptimization problems, next step's upload and extraction
    overlapped; `n_groups` such batches on their own threads so that one group's transfers overlap another's kernels, and
    `n_procs` such driver processes side by side on the GPU - calls of one process into the HIP runtime serialise, two
    processes reach 1.4x the throughput of one).  The driver runs as a child process on this rank's GPU; images are in page-locked host memory when the
    clock starts, so the figure includes every PCIe transfer of the tracking loop but no disk I/O."""
    import shutil
    import subprocess
    import tempfile
    from pointslot_amd import parallel, sequence
    exe = os.path.join(ROOT, "build", "stereo_kitti_batch")
    if not os.path.exists(exe):
        try:
            import __graft_entry__
            __graft_entry__.build_examples()
        except Exception:   # noqa: BLE001  (reported below as a failed leg)
            pass
    tmp = tempfile.mkdtemp(prefix="ps_lockstep_%d_" % rank)
    st, err, failure = None, float("inf"), None
    try:
        # a failure of this leg must not take the bench line down, and every rank must still reach the reductions below
        dirs, truth = [], []
        for k in range(n_distinct):
            seq = sequence.generate(n_frames=n_frames, seed=40 + 16 * rank + k, step=0.05 + 0.01 * k)
            d = os.path.join(tmp, "%04d" % k)
            sequence.write_pgm(d, seq)
            dirs.append(d); truth.append(seq["twc"][:, :, 3])
        cmd = [exe, "--device", str(local_rank), "--groups", str(n_groups)] + [dirs[i % n_distinct] for i in range(n_sequences)]
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(n_procs)]
        sts = []
        for pr in procs:
            so, se = pr.communicate(timeout=600)
            if pr.returncode != 0:
                raise RuntimeError("stereo_kitti_batch failed: " + so[-400:] + se[-400:])
            sts.append(json.loads(so.strip().splitlines()[-1]))
        st = dict(sts[0])
        # all processes together: frames of the timed steps over the span from the first start to the last end
        span = max(x["unix_end"] for x in sts) - min(x["unix_start"] for x in sts)
        st["wall_ms_timed_steps"] = span * 1e3
        st["untracked_frames"] = sum(x["untracked_frames"] for x in sts)
        err = 0.0
        for d, tw in zip(dirs, truth):
            traj = np.loadtxt(os.path.join(d, "CameraTrajectoryBatch.txt")).reshape(-1, 12)
            err = max(err, float(np.abs(traj[:, [3, 7, 11]] - tw).max())) if len(traj) == n_frames else float("inf")
    except Exception as e:   # noqa: BLE001
        failure = "%s: %s" % (type(e).__name__, e)
        st = None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # wall clock of the timed steps (all groups, common start) per step of n_sequences frames
    ms = parallel.max_over_ranks(dist, st["wall_ms_timed_steps"] / max(st["timed_steps"], 1) if st else float("inf"), RED_DEV)
    untracked = parallel.max_over_ranks(dist, st["untracked_frames"] if st else -1, RED_DEV)
    err = parallel.max_over_ranks(dist, err, RED_DEV)
    out = {"workload": "BASELINE config 4: %d independent 1242x375 stereo sequences per GPU x %d frames tracked by %d driver processes x %d lockstep "
                       "groups (C++ host over the C-ABI, images in pinned host memory, all PCIe transfers included)" % (n_sequences * n_procs, n_frames, n_procs, n_groups),
           "sequences_per_gpu": n_sequences * n_procs, "processes": n_procs, "groups_per_process": n_groups, "wall_ms_per_step": ms, "tracked_frames_per_s": world * n_sequences * n_procs * 1e3 / ms,
           "untracked_frames": int(untracked), "max_abs_position_error_m": err}
    if st:
        out["ms_per_step_parts_rank0_group0"] = {k[12:]: st[k] for k in st if k.startswith("ms_per_step_")}
        out["device_rounds_per_step"] = st["device_rounds_per_step"]
    if failure:
        out["error"] = failure
    return {k: (None if isinstance(v, float) and not np.isfinite(v) else v) for k, v in out.items()}   # strict JSON


def sequence_leg(rank, world, local_rank, dist, with_cpu, n_frames=12):
    """BASELINE configs[0] / [4]: every rank tracks its own generated stereo sequence (seed = rank) through the chained hot
    path (extraction -> stereo -> projection matching -> pose optimisation, pointslot_amd.tracker), then one gather of
    the [frames][12] float32 trajectories (the only collective of the workflow)."""
    import torch
    from pointslot_amd import parallel, sequence
    from pointslot_amd.tracker import HipBackend, StereoOdometry
    seq = sequence.generate(n_frames=n_frames, seed=4 + rank)
    h, w = seq["left"][0].shape
    be = HipBackend(device=local_rank)
    vo = StereoOdometry(be, seq["K"], seq["bf"], w, h)
    vo.track(seq["left"][0], seq["right"][0])
    times = []
    for k in range(1, n_frames):
        t0 = time.perf_counter()
        vo.track(seq["left"][k], seq["right"][k])
        times.append(time.perf_counter() - t0)
    be.close()
    traj = np.zeros((n_frames, 12), np.float32)
    err = 0.0
    for k, t in enumerate(vo.trajectory):
        if t is not None:
            Rwc = t[:3, :3].T
            twc = -(Rwc @ t[:3, 3])
            traj[k] = np.concatenate([Rwc, twc[:, None]], 1).reshape(12)
            err = max(err, float(np.abs(twc - seq["twc"][k][:, 3]).max()))
    t0 = time.perf_counter()
    allt = parallel.gather_trajectories(dist, traj, RED_DEV)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3
    ms = parallel.max_over_ranks(dist, float(np.median(times)) * 1e3, RED_DEV)
    out = {"workload": "BASELINE configs[0]/[4]: %d generated 1242x375 stereo sequence(s) x %d frames, one per GPU, host-driven "
                       "tracking loop over the C-ABI (single frame in flight per sequence)" % (world, n_frames),
           "median_ms_per_frame": ms, "frames_per_s_all_sequences": world * 1e3 / ms, "tracked": int(sum(t is not None for t in vo.trajectory)),
           "max_abs_position_error_m": parallel.max_over_ranks(dist, err, RED_DEV), "trajectory_gather_ms": gather_ms,
           "gathered": [list(a.shape) for a in allt]}
    if rank == 0:
        # the same sequence through the C++ driver (examples/stereo_kitti.cpp, the reference's per-frame call structure), GPU 0
        try:
            import shutil
            import subprocess
            import tempfile
            tmp = tempfile.mkdtemp(prefix="ps_single_")
            try:
                sequence.write_pgm(tmp, seq)
                run = subprocess.run([os.path.join(ROOT, "build", "stereo_kitti"), tmp], capture_output=True, text=True, timeout=300)
                med = [l for l in run.stdout.splitlines() if l.startswith("median tracking time")]
                if run.returncode == 0 and med:
                    out["cpp_driver_median_ms_per_frame"] = float(med[0].split(":")[1].split()[0])
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        except Exception as e:   # noqa: BLE001
            out["cpp_driver_error"] = "%s: %s" % (type(e).__name__, e)
    if with_cpu and rank == 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_backend import OracleBackend
        voc = StereoOdometry(OracleBackend(), seq["K"], seq["bf"], w, h)
        t0 = time.perf_counter()
        for k in range(4):
            voc.track(seq["left"][k], seq["right"][k])
        out["cpu_port_ms_per_frame_1core"] = (time.perf_counter() - t0) * 1e3 / 4
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per step per GPU")
    ap.add_argument("--cpu-pairs", type=int, default=48, help="stereo pairs timed on the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the pose-optimisation / object-BA legs")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: this process only starts N ranks (fresh children, one per GPU)
    # and forwards rank 0's JSON line; it never touches the GPU itself.  PS_BENCH_LAUNCH_ONLY=1 (CPU test of this path):
    # every rank prints what it was started with and exits before anything imports torch.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from pointslot_amd import parallel
        sys.exit(parallel.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: %d rank(s) were started (WORLD_SIZE) but --gpus is %d" % (world, args.gpus))
    if os.environ.get("PS_BENCH_LAUNCH_ONLY") == "1":
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": world, "rank": rank, "local_rank": local_rank,
                              "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}))
        if os.environ.get("PS_BENCH_LAUNCH_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        return
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # PS_BENCH_SHARE_GPU=1 (developer switch): all ranks on GPU 0 over gloo, to exercise the multi-process path on a 1-GPU box
    share = os.environ.get("PS_BENCH_SHARE_GPU") == "1" and world > 1
    if share:
        local_rank = 0
    global RED_DEV
    RED_DEV = "cpu" if share else "cuda"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pointslot_amd import synth
    from pointslot_amd.extractor import ORBextractor

    # every rank owns its own batch (distinct seeds): weak scaling, no exchange inside the step
    batch = synth.stereo_batch(args.pairs, seed=0x51070002 + 1000 * rank, w=IMG_W, h=IMG_H)
    nimg = batch.shape[0]
    d_imgs = torch.from_numpy(batch).cuda()
    ex = ORBextractor(NFEAT, 1.2, 8, 20, 5, max_batch=nimg, device=local_rank)

    def step():
        ex.extract_batch_device(d_imgs.data_ptr(), nimg, IMG_W, IMG_H, IMG_W, IMG_W * IMG_H)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ex.enable_stage_timing(True)      # HIP events on the stream the kernels run on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    stage_ms = ex.stage_times()
    ex.enable_stage_timing(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=RED_DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # SURVEY 8f-1 (next row): stereo matching of the batch's pairs on the device-resident pyramids / descriptors
    bf, fxc = 384.38148, 721.5377
    ex.stereo_match_batch(args.pairs, bf / fxc, bf)
    barrier()
    t1 = time.perf_counter()
    for _ in range(5):
        ex.stereo_match_batch(args.pairs, bf / fxc, bf)
    barrier()
    stereo_ms = (time.perf_counter() - t1) / 5 * 1e3
    ur, dp, kept = ex.stereo_fetch(0)
    assert kept > 200 and (ur >= 0).sum() == kept

    # sanity: the timed work produced keypoints (outputs stay in HBM; fetch one image)
    kps, desc = ex.fetch(0)
    assert len(kps) >= NFEAT // 2 and desc.shape == (len(kps), 32)

    secondary = None if args.no_secondary else secondary_metrics(rank, world, local_rank, dist, not args.no_cpu)

    if rank == 0:
        total_pairs = args.pairs * world * args.steps
        value = total_pairs / dt
        # dominant KERNEL: the level stage is 8 dependent launches, every other stage is one launch
        single = [k for k in stage_ms if k in ("orb_fast_cells", "orb_quadtree", "orb_blur", "orb_describe") and ALGO_BYTES_PER_IMAGE[k] > 0]
        dom = max(single, key=lambda k: stage_ms[k])
        dom_ms = stage_ms[dom]
        algo = ALGO_BYTES_PER_IMAGE[dom] * nimg
        achieved = algo / (dom_ms * 1e-3) / 1e9
        out = {
            "metric": "tracked frames/sec KITTI stereo 1242x375 (ORB front-end)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ORBextractor 8-level pyramid on 1242x375 stereo pairs, "
                                   "2000 keypoints + 256-bit rBRIEF per image",
                       "pairs_per_step_per_gpu": args.pairs, "images_per_step_per_gpu": nimg,
                       "parallelism": "images sharded over %d GPU(s), no collective in the data path" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, nimg),
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": dom_ms, "issue": pmc_valu_issue(dom, nimg)},
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
        }
        if secondary is not None:
            secondary["stereo_matching"] = {"workload": "Frame::ComputeStereoMatches on the step's %d pairs (SURVEY 8f-1), device-resident" % args.pairs,
                                            "wall_ms_per_batch": stereo_ms, "pairs_per_s": args.pairs / (stereo_ms * 1e-3),
                                            "matches_pair0": int(kept)}
            out["secondary_metrics"] = secondary
            out["metric_ba"] = {"metric": "ms/iter 50-KF object BA (8 objects)", "value": secondary["object_ba"]["ms_per_iter"],
                                "unit": "ms", "higher_is_better": False}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(batch, args.cpu_pairs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
