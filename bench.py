#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on N MI355X of one node.

Metric (BASELINE.json): tracked frames/sec on KITTI-tracking-shaped stereo 1242x375.  A "step" = one pass of the hot path
over one batch: ONE stereo frame of each of the S independent sequences a GPU tracks in lockstep — ORB extraction of 2 S
images (8-level pyramid + blur, FAST, quadtree, rBRIEF; BASELINE configs[1] per image), ComputeStereoMatches,
SearchByProjection(cur, last) -> PoseOptimization -> SearchLocalPoints / SearchByProjection(F, points) -> PoseOptimization
(configs[2] per frame), all queued on the device by the lockstep tracker (ps_tracker_*) with the images already resident in
HBM when the clock starts.  `value` = frames that went through that chain per second over all ranks (weak scaling: every
rank owns its own sequences; the path shards by sequence, no data-path collective).  Rank 0 prints ONE JSON line.

  python bench.py --gpus N --steps K --warmup W          (starts its own N ranks when no launcher did)
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
    "orb_fast_cells": PADDED_PX,                      # FAST reads the padded pyramid once
    "orb_quadtree": 0,                                # candidate lists only (not in the pixel budget)
    "orb_describe": LEVEL_PX + NFEAT * (32 + 28),     # gather (upper bound) + outputs
}
ALGO_BYTES_PER_FRAME_POSE_ITER = 2000 * 29 + 224      # pose-opt, per LM iteration per frame
ALGO_FLOP_PER_OBJECT_BA_ITER = 103e6                  # object BA, per LM iteration per object (P=50, L=300, E=15000)
RED_DEV = "cuda"              # device of the tensors used for cross-rank reductions
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# FP64 on the VECTOR ALUs (the pose-only optimisers: dependent scalar FP64 chains per edge, no matrix shape): measured issue rate of v_fma_f64
# with 8 waves per SIMD on every CU, 4.82 cycles per wave-instruction and SIMD (profiles/r05_valu_rate.txt, tools/ubench/valu_rate.hip) =
# 64 lanes x 2 flop / 4.82 cycles x 1024 SIMDs x 2.3 GHz (the clock that run assumed)
FP64_VALU_PEAK_TFLOPS = 64 * 2 / 4.82 * 1024 * 2.3e9 / 1e12
POSE_FLOP_PER_EDGE_ITER = 240.0    # errors + linearisation + errors of one LM iteration
POSE_FLOP_PER_EDGE_TRIAL = 45.0    # one more error pass (map, project, chi2, Huber) for every damping trial beyond an iteration's first
PROFILE_ROUND = "r06"   # the committed counter tables the line reads its traffic figures from (tools/reproduce_profiles.sh)


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def legs_pmc(leg, prefixes):
    """HBM bytes per launch SEQUENCE of a leg outside the headline step, from the committed counter passes of tools/prof_legs.sh
    (profiles/<round>_legs_pmc.json: rocprofv3 --kernel-trace + separate --pmc passes of the leg's tool script, TCC_EA0_RDREQ size
    classes / TCC_EA0_WRREQ(_64B)): the sum over the kernels whose name starts with one of `prefixes` of (read + write bytes per
    launch), i.e. one of each - a damping trial of the object BA, one call of the matcher / optimiser.  (bytes, per-kernel mean us,
    source) or (None, None, None)."""
    t = _profile_json(PROFILE_ROUND + "_legs_pmc.json")
    if not t or leg not in t.get("legs", {}):
        return None, None, None
    tot, found, us = 0.0, 0, {}
    for k, v in t["legs"][leg].items():
        if any(k.startswith(p) for p in prefixes) and "read_bytes_per_launch" in v:
            tot += v["read_bytes_per_launch"] + v.get("write_bytes_per_launch", 0.0)
            us[k] = v["mean_us"]
            found += 1
    return (tot, us, "profiles/%s_legs_pmc.json (leg %s), kernel table profiles/%s_%s_kernel_stats.csv" % (PROFILE_ROUND, leg, PROFILE_ROUND, leg)) if found else (None, None, None)


def pmc_traffic(kernel, nimg):
    """HBM bytes per step of `kernel` (a name, or a tuple of names whose bytes are added) from the committed PMC measurement.  r03
    (tools/pmc_step.sh): TCC_EA0_RDREQ in its 32 / 64 / 128 B size classes and TCC_EA0_WRREQ(_64B), separate passes over this bench's own
    step - bytes = requests x their size, no calibration factor (FETCH_SIZE = RDREQ x 64 B reports half of a 128-byte request).  Older
    files (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE calibrated on known-byte kernels) are read when r03's has no entry.  None when no
    measurement for this launch shape is on file.  PMC counters need rocprofv3 around the process, so this figure is read from
    profiles/, not measured in this run: `traffic_source`."""
    names = kernel if isinstance(kernel, tuple) else (kernel,)
    for name in (PROFILE_ROUND + "_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        t = _profile_json(name)
        if not t or t.get("images_per_launch") != nimg:
            continue
        tot, found = 0.0, 0
        for k, v in t.get("kernels", {}).items():
            if k.split("<")[0] in names:
                b = v.get("hbm_bytes_per_step", v.get("hbm_bytes_per_launch", v.get("hbm_bytes_per_launch_fetch_doubled")))
                if b is not None:
                    tot += b
                    found += 1
        if found:
            return tot, "profiles/" + name
    return None, None


def unit_busy(kernels):
    """The roofline of a stage that SURVEY 8d does not price in bytes (latency / issue bound kernels): how busy the busiest execution unit of
    its kernels was - vector ALUs, the CU's scalar unit, LDS - as a fraction of the kernels' own cycles, weighted by those cycles, from the
    committed counter passes of tools/valu_busy.sh (profiles/<round>_unit_busy.json: one lockstep group of 512 sequences, kernels one after
    the other).  1.0 would be a unit issuing every cycle on every CU.  Counters need rocprofv3 around the process, so the figure is read from
    profiles/, not measured in this run.  -> dict or None."""
    for rnd in (PROFILE_ROUND, "r05"):
        t = _profile_json(rnd + "_unit_busy.json")
        if t:
            break
    else:
        return None
    cyc, acc = 0.0, {"valu": 0.0, "salu": 0.0, "lds": 0.0}
    for k, v in t.get("kernels", {}).items():
        if any(k == n or k.startswith(n) for n in kernels):
            c = float(v.get("cycles", 0.0))
            cyc += c
            for u in acc:
                acc[u] += c * float(v.get(u, 0.0))
    if cyc <= 0:
        return None
    busy = {u: acc[u] / cyc for u in acc}
    top = max(busy, key=busy.get)
    return {"busy": {u: round(x, 4) for u, x in busy.items()}, "unit": top, "frac": busy[top], "source": "profiles/%s_unit_busy.json" % rnd}


def pmc_valu_issue(kernel, nimg, ms_per_step=None):
    """Share of the kernel's time per step that VALU issue alone accounts for (waves x VALU instructions x 4 cycles / 1024 SIMDs /
    clock), from the committed PMC pass (tools/pmc_step.sh: SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_WAVES, own run).  Explains a low HBM
    fraction: the kernel is bound by instruction issue, not by memory.  None when no measurement for this launch shape is on file."""
    for name in (PROFILE_ROUND + "_valu_issue.json", "r02_valu_issue.json", "r01_valu_issue.json"):
        t = _profile_json(name)
        if not t or t.get("images_per_launch") != nimg:
            continue
        for k, v in t.get("kernels", {}).items():
            if k.split("<")[0] == kernel:
                if "valu_issue_ms_per_step" in v:
                    return {"valu_issue_ms_per_step": v["valu_issue_ms_per_step"],
                            "valu_issue_share": (v["valu_issue_ms_per_step"] / ms_per_step) if ms_per_step else None,
                            "valu_instructions_per_wave": v.get("valu_per_wave"), "salu_instructions_per_wave": v.get("salu_per_wave"),
                            "waves_per_step": v.get("waves_per_step"), "source": "profiles/" + name}
                return {"valu_issue_share": v.get("valu_issue_share"), "valu_instructions_per_wave": v.get("valu_per_wave"),
                        "waves_per_launch": v.get("waves_per_launch"), "source": "profiles/" + name}
    return None


def _seq_cache_dir():
    """PS_SEQ_CACHE, default a per-user directory under /tmp (mode 0700, owned by this uid: the files are pickles, and a directory another
    user could have prepared is not read from); empty string = off"""
    root = os.environ.get("PS_SEQ_CACHE")
    if root == "":
        return None
    if root is None:
        root = os.path.join(os.environ.get("XDG_CACHE_HOME") or "/tmp", "pointslot_seq_cache_%d" % os.getuid())
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
        st = os.stat(root)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            return None
    except OSError:
        return None
    return root


def _seq_cache_path(job):
    """rendered sequences are kept on disk, keyed by the job, the generator's source and this file's render parameters: a second run on the
    same host, the other ranks' identical jobs and the secondary legs do not render again"""
    root = _seq_cache_dir()
    if not root:
        return None
    import hashlib
    import inspect
    hsh = hashlib.sha256(repr(tuple(job)).encode())
    for name in ("sequence.py", "object_tracker.py"):
        with open(os.path.join(ROOT, "pointslot_amd", name), "rb") as f:
            hsh.update(f.read())
    hsh.update(inspect.getsource(_render_one).encode())
    hsh.update(inspect.getsource(_config5_sequence).encode())
    return os.path.join(root, hsh.hexdigest()[:24] + ".pkl")


def _load_cached(job):
    import pickle
    path = _seq_cache_path(job)
    if path and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return pickle.load(f)
        except Exception:   # noqa: BLE001  (a torn or stale file: render again)
            pass
    return None


def _make_one(job):
    import pickle
    path = _seq_cache_path(job)
    q = _load_cached(job)
    if q is not None:
        return q
    q = _render_one(job)
    if path:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                pickle.dump(q, f, protocol=4)
            os.replace(tmp, path)          # atomic: a reader sees the whole file or none
        except OSError:
            pass
    return q


def _render_one(job):
    scene, n_frames, seed, k, texture = job[:5]
    n_objects = job[5] if len(job) > 5 else 2
    from pointslot_amd import sequence
    tex = sequence.kitti_texture() if texture == "kitti" else None
    if scene == "drive":
        q = sequence.generate_drive(n_frames=n_frames, seed=seed, speed=0.55 + 0.02 * (k % 16), yaw_rate_deg=0.3 + 0.05 * (k % 9), texture=tex, n_objects=n_objects)
    else:
        q = sequence.generate(n_frames=n_frames, seed=seed, step=0.05 + 0.01 * (k % 4), texture=tex)
    q["masks"] = np.stack([sequence.frame_mask(q, i) for i in range(n_frames)])
    q["dets"] = [sequence.frame_detections(q, i) for i in range(n_frames)]
    del q["seg"]
    return q


def make_sequences(rank, n_frames, n_distinct, texture, scene="drive", n_objects=2):
    """n_distinct generated stereo sequences (seeds differ per rank), rendered by a pool of fresh processes (the ray-cast drive scene
    costs about 0.3 s of numpy per frame)."""
    jobs = [(scene, n_frames, 40 + 64 * rank + k, k, texture, n_objects) for k in range(n_distinct)]
    if n_distinct <= 2:
        return [_make_one(j) for j in jobs]
    # what an earlier run on this host left in the cache is read here, in this process: a run under rocprofv3 --pmc (the profiler has the
    # GPU open before python starts, so it may not start child processes) then needs no pool at all
    have = [_load_cached(j) for j in jobs]
    if all(q is not None for q in have):
        return have
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    # sized from the CPUs this process is GRANTED (affinity cut by the cgroup quota: the GPU boxes show 256 logical CPUs and grant 16),
    # shared between the ranks of the node
    world = int(os.environ.get("WORLD_SIZE", "1"))
    todo = [j for j, q in zip(jobs, have) if q is None]
    with ProcessPoolExecutor(max_workers=min(len(todo), max(1, _cpu_quota() // max(world, 1)), 48), mp_context=mp.get_context("spawn")) as pool:
        made = iter(list(pool.map(_make_one, todo)))
    return [q if q is not None else next(made) for q in have]


MAX_OBJECTS = 8     # detections per frame the bench's trackers are created for (the generated scenes carry 2 or 6; the device chain serves up to 16)


def tracking_leg(rank, local_rank, texture, steps, warmup, n_seq, n_groups, barrier, scene="drive", n_distinct=32, objects=True, seqs=None, n_objects=2, max_objects=None):
    """The headline loop: `n_seq` sequences per GPU in `n_groups` lockstep groups (one ps_tracker and one stream each), images - and
    with `objects` the instance masks and the detections (SLOT.MODE 4 inputs) - of all frames resident in HBM.  Every step is one
    frame of every sequence through the camera chain and, with `objects`, the object chain behind it.  Returns the timing, the
    per-stage HIP-event times of group 0 and the checks on what was tracked."""
    import torch
    from pointslot_amd.tracker_device import LockstepTracker, pack_detections
    max_objects = MAX_OBJECTS if max_objects is None else max_objects
    n_frames = warmup + steps
    n_distinct = min(n_distinct, n_seq)
    if seqs is None:
        seqs = make_sequences(rank, n_frames, n_distinct, texture, scene, n_objects)
    h, w = seqs[0]["left"][0].shape
    per_group = n_seq // n_groups
    base = torch.from_numpy(np.stack([np.stack([q["left"][:n_frames], q["right"][:n_frames]], 1) for q in seqs], 1)).cuda()   # [n, nd, 2, h, w]
    # sequence j of group g shows generated sequence (g * per_group + j) % n_distinct
    imgs, masks, dets = [], [], []
    if objects:
        mbase = torch.from_numpy(np.stack([q["masks"][:n_frames] for q in seqs], 1)).cuda()                                  # [n, nd, h, w]
        dbase = np.stack([pack_detections([q["dets"][i] for q in seqs], max_objects) for i in range(n_frames)])               # [n, nd, K]
    for g in range(n_groups):
        idx = (torch.arange(per_group, device="cuda") + g * per_group) % n_distinct
        imgs.append(base[:, idx].contiguous())                                                         # [n, per_group, 2, h, w]
        if objects:
            masks.append(mbase[:, idx].contiguous())
            dets.append(torch.from_numpy(np.ascontiguousarray(dbase[:, idx.cpu().numpy()]).view(np.uint8)).cuda())
    del base
    if objects:
        del mbase
    trks = []
    try:
        for _ in range(n_groups):
            trks.append(LockstepTracker(per_group, seqs[0]["K"], seqs[0]["bf"], w, h, max_steps=n_frames, device=local_rank, max_objects=max_objects if objects else 0,
                                        max_map_objects=max(8, max_objects) if objects else 0))

        def step(i):
            for g, t in enumerate(trks):
                if objects:
                    t.step_slot_device(imgs[g][i].data_ptr(), masks[g][i].data_ptr(), dets[g][i].data_ptr())
                else:
                    t.step_device(imgs[g][i].data_ptr())

        def sync():
            for t in trks:
                t.sync()

        for i in range(warmup):
            step(i)
        sync()
        barrier()
        for t in trks:
            t.enable_stage_timing(True)      # HIP events on the stream the kernels run on
        t0 = time.perf_counter()
        for i in range(warmup, n_frames):
            step(i)
        sync()
        barrier()
        dt = time.perf_counter() - t0      # the contract's bracket: barrier + synchronize on both sides of the K steps
        stage = trks[0].stage_times()
        out = {"dt": dt, "stage_ms_group0": stage, "frames_per_step_per_gpu": per_group * n_groups, "images_per_launch": 2 * per_group,
               "h": h, "w": w, "n_distinct": n_distinct}
        # every trajectory against the ground truth of the generator, every frame's tracked flag
        err, untracked, tracked_timed, overflowed = 0.0, 0, 0, 0
        tcw0 = st0 = obj0 = None
        ob = {"detections": 0, "with_object": 0, "track_ok": 0, "max_abs_centre_error_m": 0.0, "reinit": 0, "dsd_tested": 0, "dsd_dynamic": 0}
        for g, t in enumerate(trks):
            tcw, st = t.fetch()
            untracked += int((st["tracked"] == 0).sum())
            # a frame whose search windows overflowed the candidate store is not the reference's result: it does not count as tracked
            overflowed += int((st["overflowed"] != 0).sum())
            tracked_timed += int(((st["tracked"][warmup:] != 0) & (st["overflowed"][warmup:] == 0)).sum())
            if g == 0:
                tcw0, st0 = tcw, st
            R = tcw[:, :, :3, :3]
            twc = -np.einsum("nsji,nsj->nsi", R, tcw[:, :, :3, 3])
            for j in range(per_group):
                truth = seqs[(g * per_group + j) % n_distinct]["twc"][:n_frames, :, 3]
                ok = st["tracked"][:, j] != 0
                if ok.any():
                    err = max(err, float(np.abs(twc[ok, j] - truth[ok]).max()))
            if objects:
                o = t.fetch_objects()
                if g == 0:
                    obj0 = o
                live = o["id"] >= 0
                ob["detections"] += int(live[2:].sum()); ob["with_object"] += int((o["tracked"][2:] != 0).sum())
                ob["track_ok"] += int((o["track_ok"][2:] != 0).sum()); ob["reinit"] += int(o["reinit"].sum())
                ran = (o["dyn_n_mono"] + o["dyn_n_stereo"]) > 0          # DynamicStaticDiscrimination's reprojection test ran (depth / centre gates passed)
                ob["dsd_tested"] += int(ran.sum()); ob["dsd_dynamic"] += int((ran & (o["dynamic"] != 0)).sum())
                if scene == "drive":
                    # the generator's even-numbered objects drive ahead, the odd ones stand at the roadside (sequence.generate_drive): where the test
                    # ran, its verdict against that
                    moving = (o["id"] % 2 == 0)
                    ob["dsd_agrees_with_generator"] = ob.get("dsd_agrees_with_generator", 0) + int((ran & ((o["dynamic"] != 0) == moving)).sum())
                for j in range(min(per_group, n_distinct)):                  # the distinct sequences once: cuboid centres against the labels
                    q = seqs[(g * per_group + j) % n_distinct]
                    for i in range(2, n_frames):
                        for k, d in enumerate(q["dets"][i]):
                            if o[i, j, k]["track_ok"]:
                                ob["max_abs_centre_error_m"] = max(ob["max_abs_centre_error_m"], float(np.abs(o[i, j, k]["tco"][:3] - d["pose7"][:3]).max()))
        out.update(max_abs_position_error_m=err, untracked_frames=untracked, tracked_frames_timed=tracked_timed, overflowed_frames=overflowed, seqs=seqs, tcw_group0=tcw0, stats_group0=st0, obj_group0=obj0,
                   objects=ob if objects else None)
    finally:       # a leg abandoned by an exception must not leave its arenas beside the next leg
        for t in trks:
            t.close()
        del imgs, masks, dets
        torch.cuda.empty_cache()
    return out


class _ThreadedOracle:
    """The CPU checker behind the chain with the reference's thread model (src/Frame.cc:709-714,2648-2651): the two ORBextractor
    calls on two threads, the two cv::ORB detectors of ExtractObjORB on two threads (the oracle calls release the GIL)."""

    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        from oracle_backend import OracleBackend
        self.o = OracleBackend()
        self.lib = oracle_lib
        self.cv2 = oracle_lib.OracleCvORB(1000, 1.2, 8, 19, 20)
        self.scale_factors, self.inv_level_sigma2 = self.o.scale_factors, self.o.inv_level_sigma2

    def extract_stereo(self, left, right, mb, mbf):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as pool:
            a = pool.submit(self.o.left.run, left); b = pool.submit(self.o.right.run, right)
            kps, desc = a.result(); b.result()
        _, ur, dp = self.lib.stereo_match(self.o.left, self.o.right, mb, mbf)
        return kps, desc, ur, dp

    def extract_objects(self, left, right, ml, mr, mb, mbf):
        from concurrent.futures import ThreadPoolExecutor
        if not hasattr(self.o, "cv"):
            self.o.cv = self.lib.OracleCvORB(1000, 1.2, 8, 19, 20)
        with ThreadPoolExecutor(2) as pool:
            a = pool.submit(self.o.cv.run, left, ml); b = pool.submit(self.cv2.run, right, mr)
            (kl, dl), (kr, dr) = a.result(), b.result()
        if len(kl) == 0:
            return kl, dl, np.zeros(0, np.float32), np.zeros(0, np.float32)
        _, ur, dp = self.lib.stereo_match_keys(self.o.left, self.o.right, kl, dl, kr, dr, mb, mbf)
        return kl, dl, ur, dp

    def __getattr__(self, name):
        return getattr(self.o, name)


def _cpu_chain(q, n, backend, objects=True):
    """n frames of sequence q through the CPU restatement of the chain; returns (StereoOdometry, seconds)"""
    from pointslot_amd.tracker import StereoOdometry
    h, w = q["left"][0].shape
    vo = StereoOdometry(backend, q["K"], q["bf"], w, h)
    t0 = time.perf_counter()
    for i in range(n):
        if objects:
            vo.track(q["left"][i], q["right"][i], q["masks"][i], q["dets"][i])
        else:
            vo.track(q["left"][i], q["right"][i])
    return vo, time.perf_counter() - t0


def _cpu_worker(job):
    """one process of the all-cores baseline: a slice of a sequence from the shared file, through the chain, single-threaded"""
    path, k, n, objects = job
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleBackend
    with open(path, "rb") as f:
        seqs = pickle.load(f)
    _, dt = _cpu_chain(seqs[k % len(seqs)], n, OracleBackend(), objects)
    return n, dt


_ONE_THREAD_ENV = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "VECLIB_MAXIMUM_THREADS")


def _physical_cores():
    """physical cores of the host (distinct (package, core) pairs of the logical CPUs)"""
    try:
        seen = set()
        base = "/sys/devices/system/cpu"
        for d in os.listdir(base):
            if d.startswith("cpu") and d[3:].isdigit():
                t = os.path.join(base, d, "topology")
                with open(os.path.join(t, "physical_package_id")) as f:
                    pk = f.read().strip()
                with open(os.path.join(t, "core_id")) as f:
                    seen.add((pk, f.read().strip()))
        return len(seen) or None
    except OSError:
        return None


def _cpu_quota():
    """CPUs this process may actually use: the scheduler affinity, cut by the cgroup's CPU-time quota (cpu.max) when there is one -
    the GPU boxes of this pool show 256 logical CPUs and grant 16 CPUs of time"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            with open(path) as f:
                q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per))))
        except (OSError, ValueError):
            pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


def _cpp_camera_chain_baseline(q, n=8):
    """tests/cpp/odo_oracle_driver (pointslot_amd/host/StereoOdometry.h's OdoSequence with oracle/liboracle.so serving its requests) on the
    first n frames of a generated sequence: the camera chain on all keypoints, one thread, image decode outside the timer."""
    import subprocess
    import tempfile
    from pointslot_amd import sequence
    exe = os.path.join(ROOT, "tests", "cpp", "odo_oracle_driver")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".cpp"):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "pointslot_amd", "host"), "-I", os.path.join(ROOT, "include"), exe + ".cpp", "-o", exe,
                               "-L", os.path.join(ROOT, "oracle"), "-loracle", "-L", os.path.join(ROOT, "pointslot_amd"), "-lpointslot_hip", "-pthread",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        sub = {k: (v[:n] if k in ("left", "right", "twc", "boxes", "masks", "dets") and hasattr(v, "__len__") else v) for k, v in q.items()}
        sequence.write_pgm(os.path.join(d, "0000"), sub)
        out = subprocess.run([exe, os.path.join(d, "0000")], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-300:])
    line = [l for l in out.stdout.splitlines() if l.startswith("timing:")][-1].split()
    frames, secs = int(line[1]), float(line[3])
    tracked = sum(1 for l in out.stdout.splitlines() if l.startswith("frame") and ": ok" in l)
    return {"value": frames / secs, "unit": "frames/s", "cores": 1, "kind": "port", "tracked_frames": tracked,
            "sample": "%d frames of the first generated sequence through the C++ host loop (OdoSequence) over the CPU checker: camera chain on all "
                      "keypoints (no object chain), one thread, no Python in the loop" % frames}


def cpu_tracking_baseline(seqs, tcw_gpu, obj_gpu, objects, budget_s=10.0):
    """The CPU restatement of the same loop (tests/oracle_backend.py over oracle/liboracle.so) on the sequences the GPU just tracked:
    (a) one core, the reference's per-frame call structure - the `cpu_baseline` and the parity spot check of the timed run (sequence j of
    group 0 shows generated sequence j); (b) the reference's thread model (two extraction threads + two object-detector threads);
    (c) all host cores, one independent sequence slice per process."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleBackend
    frames, spent, worst, checked, obj_checked, obj_bad = 0, 0.0, 0.0, 0, 0, 0
    k = 0
    n_per = min(len(seqs[0]["left"]), 8)
    while spent < budget_s and k < len(seqs):
        q = seqs[k]
        vo, dt = _cpu_chain(q, n_per, OracleBackend(), objects)
        spent += dt; frames += n_per
        if k < tcw_gpu.shape[1]:
            for i in range(n_per):
                a, b = vo.trajectory[i], tcw_gpu[i, k]
                worst = max(worst, float(np.abs(b).max()) if a is None else float(np.abs(a - b).max()))   # no pose: the GPU row is zeros
                checked += 1
                if objects and obj_gpu is not None:
                    for j, o in enumerate(vo.objects.stats[i]["objects"]):
                        d = obj_gpu[i, k, j]
                        same = (int(d["n"]) == o["n"] and int(d["tracked"]) == int(o["tracked"]) and int(d["bf_matches"]) == o["bf_matches"]
                                and int(d["lm_matches"]) == o["lm_matches"] and int(d["inliers"]) == o["inliers"]
                                and int(d["dynamic"]) == int(o["dynamic"]) and (int(d["dyn_n_mono"]), int(d["dyn_n_stereo"])) == tuple(o["dyn_n"]))
                        obj_checked += 1; obj_bad += 0 if same else 1
        k += 1
    one = {"value": frames / spent, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames (the first %d of %d of the run's sequences) through the CPU restatement of the same per-frame chain%s (oracle/*.cpp, -O3 "
                     "-march=x86-64-v3, one thread); host has %d cores" % (frames, n_per, k, " incl. the object chain" if objects else "", os.cpu_count())}
    # (b) the reference's thread model
    t_frames, t_spent = 0, 0.0
    kk = 0
    while t_spent < 0.5 * budget_s and kk < len(seqs):
        _, dt = _cpu_chain(seqs[kk], n_per, _ThreadedOracle(), objects)
        t_spent += dt; t_frames += n_per; kk += 1
    threaded = {"value": t_frames / t_spent, "unit": "frames/s", "cores": 2, "kind": "port",
                "sample": "%d frames, the reference's thread model: ExtractORB left / right on two threads, the two cv::ORB detectors of ExtractObjORB on two "
                          "threads, everything else on the tracking thread (src/Frame.cc:709-714,2648-2651)" % t_frames}
    # (c) all cores: independent sequence slices, one single-threaded process each.  The workers' numeric libraries are pinned to one
    # thread (numpy's BLAS pool otherwise starts a thread per core in every one of them), the pool is as large as the host has
    # PHYSICAL cores, it is warm before the timer starts (imports, library load, one frame), and the scaling against the one-core
    # figure is part of the result.
    allc = None
    try:
        import multiprocessing as mp
        import pickle
        import tempfile
        from concurrent.futures import ProcessPoolExecutor
        ncpu = os.cpu_count() or 1
        phys = _physical_cores() or ncpu
        quota = _cpu_quota()
        nproc = max(1, min(phys, quota, 256))
        n_all = 5
        slim = [{kk2: (q[kk2][:n_all] if kk2 in ("left", "right", "masks", "dets") else q[kk2]) for kk2 in ("left", "right", "masks", "dets", "K", "bf")} for q in seqs[:8]]
        fd, path = tempfile.mkstemp(suffix=".pkl", dir="/tmp")
        with os.fdopen(fd, "wb") as f:
            pickle.dump(slim, f, protocol=4)
        saved = {k2: os.environ.get(k2) for k2 in _ONE_THREAD_ENV}
        os.environ.update({k2: "1" for k2 in _ONE_THREAD_ENV})     # inherited by the spawned workers
        try:
            with ProcessPoolExecutor(max_workers=nproc, mp_context=mp.get_context("spawn")) as pool:
                list(pool.map(_cpu_worker, [(path, i, 1, objects) for i in range(nproc)]))          # start the workers (imports, library load)
                t0 = time.perf_counter()
                res = list(pool.map(_cpu_worker, [(path, i, n_all, objects) for i in range(nproc)]))
                wall = time.perf_counter() - t0
        finally:
            for k2, v2 in saved.items():
                if v2 is None:
                    os.environ.pop(k2, None)
                else:
                    os.environ[k2] = v2
        os.unlink(path)
        per = sorted(r[0] / r[1] for r in res)
        val = sum(r[0] for r in res) / wall
        allc = {"value": val, "unit": "frames/s", "cores": nproc, "kind": "port", "logical_cpus": ncpu, "physical_cores": phys, "cpu_quota_of_this_process": quota,
                "per_process_frames_per_s": {"min": per[0], "median": per[len(per) // 2], "max": per[-1]},
                "scaling_efficiency_vs_one_core": val / (nproc * one["value"]),
                "sample": "%d single-threaded processes (one per core this process may use: physical cores cut by the cgroup's CPU quota - r03's 192 "
                          "processes shared the 16 CPUs of time the box grants; OMP / OpenBLAS / MKL threads = 1) x %d frames, one independent sequence slice "
                          "each, pool warm before the timer; efficiency = value / (processes x the one-core figure): what is missing is the host, not the "
                          "chain - the per-process rates show whether the cores slow each other down (shared L3 / memory bandwidth / clocks under an "
                          "all-core load)" % (nproc, n_all)}
    except Exception as e:   # noqa: BLE001
        allc = {"error": "%s: %s" % (type(e).__name__, e)}
    # (d) the camera chain through the C++ host state machine with the same checker behind it: no Python in the loop
    cpp = None
    try:
        cpp = _cpp_camera_chain_baseline(seqs[0])
    except Exception as e:   # noqa: BLE001
        cpp = {"error": "%s: %s" % (type(e).__name__, e)}
    one["note"] = ("a scalar port: ~90 % of this time is oracle C++ (ORBextractor 49 ms / image, cv::ORB 36 ms / image), ~10 % Python glue; the real "
                   "OpenCV (SIMD FAST / resize / blur) would be faster - it cannot be built in this image")
    one["camera_chain_cpp_driver"] = cpp
    return one, threaded, allc, worst, checked, obj_checked, obj_bad


def orb_leg(rank, local_rank, barrier, with_cpu):
    """BASELINE configs[1] on its own: the ORB extractor over 64 synthetic stereo pairs resident in HBM (the r01 headline), with
    an in-process parity spot check of the timed batch against the CPU restatement."""
    import torch
    from pointslot_amd import synth
    from pointslot_amd.extractor import ORBextractor
    pairs = 64
    batch = synth.stereo_batch(pairs, seed=0x51070002 + 1000 * rank, w=IMG_W, h=IMG_H)
    nimg = batch.shape[0]
    d_imgs = torch.from_numpy(batch).cuda()
    ex = ORBextractor(NFEAT, 1.2, 8, 20, 5, max_batch=nimg, device=local_rank)
    for _ in range(3):
        ex.extract_batch_device(d_imgs.data_ptr(), nimg, IMG_W, IMG_H, IMG_W, IMG_W * IMG_H)
    barrier()
    ex.enable_stage_timing(True)
    t0 = time.perf_counter()
    for _ in range(10):
        ex.extract_batch_device(d_imgs.data_ptr(), nimg, IMG_W, IMG_H, IMG_W, IMG_W * IMG_H)
    barrier()
    dt = (time.perf_counter() - t0) / 10
    stage = ex.stage_times()
    ex.enable_stage_timing(False)
    bf, fxc = 384.38148, 721.5377
    ex.stereo_match_batch(pairs, bf / fxc, bf)
    barrier()
    t1 = time.perf_counter()
    for _ in range(5):
        ex.stereo_match_batch(pairs, bf / fxc, bf)
    barrier()
    stereo_ms = (time.perf_counter() - t1) / 5 * 1e3
    out = {"workload": "BASELINE configs[1]: ORBextractor 8-level pyramid on 64 synthetic 1242x375 stereo pairs in HBM, 2000 keypoints + 256-bit rBRIEF per image",
           "ms_per_step": dt * 1e3, "stereo_frames_per_s": pairs / dt, "stage_ms": {k: round(v, 5) for k, v in stage.items()},
           "stereo_matching_ms_per_64_pairs": stereo_ms}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from concurrent.futures import ThreadPoolExecutor
        from oracle_lib import OracleORB
        n_check = 8
        orcs = [OracleORB(NFEAT), OracleORB(NFEAT)]
        res = [None] * (2 * n_check)

        def run(side):
            for k in range(n_check):
                res[2 * k + side] = orcs[side].run(batch[2 * k + side])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(2) as pool:
            list(pool.map(run, [0, 1]))
        cpu_dt = time.perf_counter() - t0
        bad = 0
        for i in range(2 * n_check):
            kps, desc = ex.fetch(i)
            ko, do = res[i]
            if len(kps) != len(ko) or not np.array_equal(kps.view(np.uint8), ko.view(np.uint8)) or not np.array_equal(desc, do):
                bad += 1
        out["parity_spot"] = "green" if bad == 0 else "red"
        out["parity_spot_detail"] = "%d of the timed batch's images: keypoints and descriptors bit-exact vs the CPU restatement (%d differ)" % (2 * n_check, bad)
        out["cpu_port_stereo_frames_per_s_2_threads"] = n_check / cpu_dt
    ex.close()
    del d_imgs
    return out


def optimizer_legs(rank, world, local_rank, guard, with_cpu, fp64_peak):
    """BASELINE configs[2] and [3]: per-frame pose optimisation (batch of 64 frames x 2000 stereo edges) and the object local BA
    (8 objects x 50 keyframes x 300 points, SURVEY.md 8d's own perturbation: +-0.3 m / +-5 deg yaw / +-0.1 m).  Frames / objects
    are independent units and are sharded over the ranks; times are max-over-ranks."""
    from pointslot_amd import parallel, synth
    from pointslot_amd.optimizer import Optimizer
    opt = Optimizer(device=local_rank)
    out = {}
    mine = list(parallel.shard_units(64, world, rank))
    frames = [synth.pose_problem(0x51070003 + k) for k in mine]
    opt.PoseOptimization(frames[:1])                       # warm-up
    t0 = time.perf_counter()
    res = opt.PoseOptimization(frames)
    wall = time.perf_counter() - t0
    kern_ms = opt.last_kernel_ms()
    opt.enable_trace(True)                                 # untimed repeat with the per-iteration log: the edge passes of the timed call
    opt.PoseOptimization(frames)
    traces = [opt.get_trace(i) for i in range(len(frames))]
    opt.enable_trace(False)
    iters = sum(len(t) for t in traces)
    trials = sum(int(t[:, 2].sum()) for t in traces)
    wall = guard.max(wall)
    kern_ms = guard.max(kern_ms)
    # roofline of the persistent kernel (58 KB per frame and LM iteration in the streaming model; every damping
    # trial re-reads the edges once more): the kernel is latency-bound, the fraction says by how much
    algo = (iters + trials) * ALGO_BYTES_PER_FRAME_POSE_ITER
    out["pose_optimization"] = {"workload": "BASELINE configs[2]: 64 frames x (1 SE3 x 2000 stereo edges), 4 x 10 LM schedule",
                                "frames": 64, "kernel_ms_per_batch": kern_ms, "wall_ms_per_batch_incl_pcie": wall * 1e3,
                                "frames_per_s_kernel": 64 / (kern_ms * 1e-3), "inliers_frame0": int(res[0][0]) if res else None,
                                "lm_iterations": iters, "damping_trials": trials,
                                "roofline": {"bound": "hbm", "kernel": "pose_lm", "achieved": algo / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "traffic": legs_pmc("pose", ("pose_lm",))[0], "traffic_source": legs_pmc("pose", ("pose_lm",))[2],
                                             "algorithmic_bytes_per_launch": algo, "avg_launch_ms": kern_ms,
                                             "note": "edge passes x 58 KB (SURVEY 8d streaming model); the persistent kernel keeps its edges in L2 and is latency-bound"}}
    # ... and against what actually bounds it: FP64 on the vector ALUs.  flops of the passes the kernel ran (its own per-iteration log: SURVEY's
    # 240 flop per edge and LM iteration, 45 for every further damping trial's error pass; 2000 edges minus the ones sitting a round out are
    # not discounted: an upper bound of the work, hence of the fraction)
    flop = 2000.0 * (POSE_FLOP_PER_EDGE_ITER * iters + POSE_FLOP_PER_EDGE_TRIAL * max(trials - iters, 0))
    ach = flop / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
    ub = unit_busy(("pose_lm",))
    out["pose_optimization"]["roofline_fp64"] = {"bound": "fp64-valu", "kernel": "pose_lm", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                 "frac": ach / FP64_VALU_PEAK_TFLOPS if ach else None, "flop_per_launch": flop,
                                                 "vector_alu_busy_in_the_tracker_step": ub["busy"]["valu"] if ub else None, "unit_busy_source": ub["source"] if ub else None,
                                                 "note": "64 workgroups of 256 threads on 256 CUs: a quarter of the chip holds one wave per SIMD; the fraction is of the WHOLE chip's "
                                                         "vector FP64 issue rate (measured v_fma_f64 rate, profiles/r05_valu_rate.txt)"}
    mine = list(parallel.shard_units(8, world, rank))
    graphs = [synth.object_ba_problem(0x51070004 + j) for j in mine]
    if graphs:
        opt.ObjectLocalBundleAdjustment(graphs[:1])        # warm-up (allocations)
        r = opt.ObjectLocalBundleAdjustment(graphs)
        ms = opt.last_kernel_ms()
        iters = max(x["iterations"] for x in r)
        trials = max(x["trials"] for x in r)
        sum_trials = sum(x["trials"] for x in r)
        r1 = opt.ObjectLocalBundleAdjustment(graphs[:1])    # SURVEY.md 8d config 4 also asks for one object alone
        ms1, iters1 = opt.last_kernel_ms(), max(r1[0]["iterations"], 1)
    else:
        ms, iters, trials, sum_trials = 0.0, 1, 1, 0
        ms1, iters1 = 0.0, 1
    ms = guard.max(ms)
    iters = int(guard.max(iters))
    ba = {"workload": "BASELINE configs[3]: 8 objects x 50 ObjectKeyFrames x 300 MapObjectPoints (15 000 stereo edges each), Schur LM 5 + 10 "
                      "iterations, SURVEY 8d perturbation (+-0.3 m, +-5 deg yaw, points +-0.1 m)", "objects": 8, "gpu_ms_per_batch": ms,
          "lm_iterations": iters, "lm_trials": int(guard.max(trials)), "ms_per_iter": ms / max(iters, 1),
          "ms_per_iter_1_object": guard.max(ms1 / iters1)}
    if graphs and ms > 0:
        # every damping trial is one linearise + Schur + solve (103 MFLOP per object): FP64 rate over the batch
        flop = sum_trials * ALGO_FLOP_PER_OBJECT_BA_ITER
        ach = flop / (ms * 1e-3) / 1e12
        tr_bytes, tr_us, tr_src = legs_pmc("ba", ("ba_linearize", "ba_prep", "ba_schur", "ba_solve", "ba_update", "ba_error_k", "ba_decide"))
        flop_iter = iters * len(graphs) * ALGO_FLOP_PER_OBJECT_BA_ITER      # SURVEY 8d prices an LM ITERATION (0.82 GFLOP for 8 objects)
        ach_iter = flop_iter / (ms * 1e-3) / 1e12
        ba["roofline"] = {"bound": "mfma", "kernel": "ba_* (7 kernels per damping trial)", "achieved": ach, "peak": fp64_peak, "unit": "TFLOP/s",
                          "frac": ach / fp64_peak if fp64_peak else None, "achieved_per_iteration": ach_iter,
                          "frac_per_iteration": ach_iter / fp64_peak if fp64_peak else None,
                          "fractions": "frac prices every damping trial (one linearise / Schur / solve each, %d in this run) at SURVEY 8d's 103 MFLOP per "
                                       "object; frac_per_iteration prices the %d LM iterations only, as SURVEY 8d does" % (sum_trials, iters),
                          "traffic": tr_bytes, "traffic_unit": "HBM bytes per damping trial of the 8-object batch (one launch of each kernel)",
                          "traffic_source": tr_src, "kernel_mean_us": tr_us, "algorithmic_flop_per_batch": flop,
                          "peak_source": "v_mfma_f64_16x16x4_f64 microbenchmark measured in this run (ps_debug_mfma_f64_peak)",
                          "note": "latency-bound: one object's reduced system is 300 unknowns; see DESIGN.md section 7"}
    ba["objects_per_gpu"] = len(mine)
    ba["n_gpus"] = world   # objects are the units: 8 -> 4 / 2 / 1 per GPU at 2 / 4 / 8 GPUs, gpu_ms_per_batch is the slowest rank's
    if graphs and world == 1:
        # where the latency floor ends: the same schedule for 1, 8, 16, 32, 64 independent objects in one batch (seeds 0x51070004 + j)
        scale = {}
        many = graphs + [synth.object_ba_problem(0x51070004 + j) for j in range(len(graphs), 64)]
        for nobj in (16, 32, 64):
            rr = opt.ObjectLocalBundleAdjustment(many[:nobj])
            scale[str(nobj)] = {"gpu_ms_per_batch": opt.last_kernel_ms(), "ms_per_iter": opt.last_kernel_ms() / max(max(x["iterations"] for x in rr), 1)}
        scale["1"] = {"gpu_ms_per_batch": ms1, "ms_per_iter": ms1 / iters1}
        scale["8"] = {"gpu_ms_per_batch": ms, "ms_per_iter": ms / max(iters, 1)}
        ba["batch_scaling"] = scale
    out["object_ba"] = ba
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
    return out


VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9      # 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md chip table): one 32-bit lane-op per lane and cycle


def object_legs(local_rank, fp64_peak):
    """The hot-path kernels of the object half on their own, rank 0, kernel times from HIP
    events on the handles' streams (host buffers: the PCIe-inclusive wall time is reported beside them)."""
    import torch
    from pointslot_amd import sequence, synth
    from pointslot_amd.matcher import ORBmatcher, build_grid
    from pointslot_amd.object_orb import ORB
    from pointslot_amd.object_tracker import object_masks, right_mask
    from pointslot_amd.optimizer import Optimizer
    out = {}
    # ---- a10 SearchByBruceMatching: k objects x (1000 last-frame features against 1000 current ones) ----
    m = ORBmatcher(0.9, True, device=local_rank)
    for k in (1, 8, 64):
        probs = [synth.bruteforce_problem(900 + i, nq=1000, nt=1000) for i in range(k)]
        m.SearchByBruceMatching(probs)
        t0 = time.perf_counter()
        m.SearchByBruceMatching(probs)
        wall = time.perf_counter() - t0
        ms = m.last_kernel_ms()
        pairs = sum(len(p["q_desc"]) * len(p["t_desc"]) for p in probs)
        ops = pairs * 16                                            # 8 x (v_xor_b32 + v_bcnt_u32_b32) per pair (Nq Nt 8 xor + popc32)
        out["bruteforce_%dx1000x1000" % k] = {
            "workload": "a10: SearchByBruceMatching, %d object(s) x 1000 x 1000 descriptors, nn ratio 0.9, rotation check" % k,
            "kernel_ms": ms, "wall_ms_incl_pcie": wall * 1e3, "pairs_per_s": pairs / (ms * 1e-3),
            "roofline": {"bound": "int-alu", "kernel": "bf_topk + bf_resolve", "achieved": ops / (ms * 1e-3) / 1e12, "peak": VALU_LANE_OPS_PER_S / 1e12,
                         "unit": "T lane-op/s", "frac": ops / (ms * 1e-3) / VALU_LANE_OPS_PER_S,
                         "traffic": legs_pmc("bf", ("bf_topk", "bf_resolve"))[0], "traffic_source": legs_pmc("bf", ("bf_topk", "bf_resolve"))[2],
                         "traffic_unit": "HBM bytes per call, mean over the leg's 1 / 8 / 64-object calls",
                         "note": "integer ALU / latency bound as SURVEY 8d states: %d KB of descriptors per object stay in LDS / L2; bf_resolve is the "
                                 "order-dependent serial part" % (2000 * 32 // 1024)}}
    # ---- a13 SearchByProjection(F, nOrder, MOPs): k objects x (300 local points into 400 features) ----
    prs = []
    for i in range(32):
        sc = synth.projection_scene(700 + i, n=400, m=300)
        T = sc["train"]
        T["cell_off"], T["cell_idx"] = build_grid(T["x"], T["y"], *T["grid"])
        prs.append({"mode": "points", "object": True, "train": T, "query": sc["points_query"], "scale_factors": sc["scale_factors"], "th": 1.0})
    m8 = ORBmatcher(0.8, True, device=local_rank)
    m8.SearchByProjection(prs)
    t0 = time.perf_counter()
    m8.SearchByProjection(prs)
    wall = time.perf_counter() - t0
    out["object_search_by_projection"] = {"workload": "a13: SearchByProjection(F, nOrder, MOPs, th = 1), 32 objects x (300 local points, 400 features, radius 5 px, bbox test)",
                                          "kernel_ms": m8.last_kernel_ms(), "wall_ms_incl_pcie": wall * 1e3,
                                          "roofline": {"bound": "latency", "kernel": "pj_gather<15> + pj_resolve<15>", "achieved": None, "peak": None, "unit": None,
                                                       "frac": None, "traffic": None, "note": "a window holds a handful of candidates: dependent grid look-ups, no throughput figure in SURVEY 8d"}}
    m.close(); m8.close()
    # ---- a15 CFSE3ObjStateOptimization: frames with k objects x ~150 points, two calls per frame ----
    opt = Optimizer(device=local_rank)
    rng = np.random.default_rng(3)

    def cf_frame(seed, k):
        objs = []
        for o in range(k):
            pp = synth.pose_problem(seed * 100 + o, n=150, outlier_frac=0.15, mono_frac=0.2, valid_frac=0.8)
            Tp = pp["tcw_true"].copy(); Tp[:3, 3] += rng.uniform(-0.2, 0.2, 3)
            from pointslot_amd.optimizer import se3_from_mat4f
            objs.append({"xo": pp["xw"], "obs": pp["obs"], "inv_sigma2": pp["inv_sigma2"], "valid": pp["valid"], "pose7": se3_from_mat4f(Tp.astype(np.float32))})
        return {"objs": objs, "K": pp["K"]}
    for k in (1, 4, 10):
        frames = [cf_frame(50 + i, k) for i in range(64)]
        opt.CFSE3ObjStateOptimization(frames[:2])
        t0 = time.perf_counter()
        opt.CFSE3ObjStateOptimization(frames)
        wall = time.perf_counter() - t0
        ms = opt.last_kernel_ms()
        edges = sum(int(np.asarray(o["valid"]).sum()) for f in frames for o in f["objs"])
        out["cfse3_%d_objects" % k] = {"workload": "a15: CFSE3ObjStateOptimization, 64 frames x %d object(s) x 150 points (one graph per frame, 4 x 10 LM)" % k,
                                       "kernel_ms_per_64_calls": ms, "wall_ms_incl_pcie": wall * 1e3, "edges": edges,
                                       "roofline": {"bound": "latency", "kernel": "pose_lm (mode 1)", "achieved": None, "peak": None, "unit": None, "frac": None,
                                                    "traffic": legs_pmc("cfse3", ("pose_lm",))[0] if k == 4 else None, "traffic_source": legs_pmc("cfse3", ("pose_lm",))[2] if k == 4 else None,
                                                    "note": "one workgroup per frame, %d x 6 unknowns: the serial LM control flow dominates" % k}}
    opt.close()
    # ---- 8f-2 the object detector: 64 stereo pairs with their object masks, device resident ----
    seq = sequence.generate_drive(n_frames=2, seed=7, texture=sequence.kitti_texture())
    mk = sequence.frame_mask(seq, 1)
    ol, orr = object_masks(mk, right_mask(mk))
    nimg = 128
    d_i = torch.from_numpy(np.stack([seq["left"][1], seq["right"][1]] * (nimg // 2))).cuda()
    d_m = torch.from_numpy(np.stack([ol, orr] * (nimg // 2))).cuda()
    h, w = ol.shape
    det = ORB(device=local_rank)
    det.detect_batch_device(d_i.data_ptr(), d_m.data_ptr(), nimg, w, h)
    kps, _ = det.batch_fetch(0)
    t0 = time.perf_counter()
    for _ in range(5):
        det.detect_batch_device(d_i.data_ptr(), d_m.data_ptr(), nimg, w, h)
    det.batch_fetch(0)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    algo = nimg * 2 * w * h
    out["object_detector"] = {"workload": "8f-2: cv::ORB(1000, 1.2, 8, 19) under the object masks, %d images of %dx%d in HBM (mask coverage %.3f), %d keypoints in image 0" % (nimg, w, h, float((ol != 0).mean()), len(kps)),
                              "ms_per_batch": ms, "images_per_s": nimg / (ms * 1e-3),
                              "roofline": {"bound": "hbm", "kernel": "cvb_* (13 launches)", "achieved": algo / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_batch": algo,
                                           "note": "algorithmic bytes = image + mask read once; the work is the part of the pyramid a masked keypoint can reach: latency-bound tile kernels"}}
    det.close()
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
    g = synth.object_ba_problem(0x51070060, n_kf=10, n_pts=1200, p_vis=0.35, perturb=(0.05, 1.0, 0.03), perturb_axis="y", n_fixed_extra=6, mono_frac=0.15)
    g["pose_flags"] = (g["pose_flags"] & 1).astype(np.uint8)
    opt.ObjectLocalBundleAdjustment([g])
    r, = opt.ObjectLocalBundleAdjustment([g])
    out["local_ba"] = {"workload": "f-3: 16 keyframes (6 fixed) x 1200 points, %d edges" % len(g["e_pose"]), "gpu_ms": opt.last_kernel_ms(),
                       "lm_iterations": int(r["iterations"]), "ms_per_iter": opt.last_kernel_ms() / max(int(r["iterations"]), 1)}
    objs = [synth.dynamic_object(100 + k, n=300, moving=0.1 * k) for k in range(8)]
    opt.DynamicStaticDiscrimination(objs)
    t0 = time.perf_counter()
    for _ in range(10):
        opt.DynamicStaticDiscrimination(objs)
    out["dynamic_static_discrimination"] = {"workload": "f-4: 8 detections x 300 object points per call", "wall_ms_per_call": (time.perf_counter() - t0) * 100}
    opt.close()
    m = ORBmatcher(0.6, True, device=local_rank)
    rng = np.random.default_rng(7)
    lists = [rng.integers(0, 256, (int(n), 32), dtype=np.uint8) for n in rng.integers(2, 51, 2400)]
    m.ComputeDistinctiveDescriptors(lists)
    t0 = time.perf_counter()
    m.ComputeDistinctiveDescriptors(lists)
    dt = time.perf_counter() - t0
    out["distinctive_descriptors"] = {"workload": "f-4: 2400 map points, 2..50 observations each (%d descriptors)" % sum(len(x) for x in lists),
                                      "wall_ms_per_call": dt * 1e3, "points_per_s": len(lists) / dt}
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


def pcie_leg(rank, world, local_rank, guard, seqs, n_seq, n_groups, barrier):
    """The same lockstep loop with the images in page-locked HOST memory (ps_tracker_step): every frame's 2 x 0.47 MB cross PCIe
    inside the timed region.  This is the rate a caller with host buffers sees; it is never `value`."""
    from pointslot_amd import parallel
    from pointslot_amd._lib import PinnedBuffer
    from pointslot_amd.tracker_device import LockstepTracker
    n = min(len(seqs[0]["left"]), 8)
    h, w = seqs[0]["left"][0].shape
    per_group, nd = n_seq // n_groups, len(seqs)
    pitch = h * w
    # one copy of every frame of every tracked sequence, laid out as a capture pipeline would hand it over: per frame, per group,
    # the group's stereo pairs back to back (one transfer per group and step)
    pin = PinnedBuffer(n * n_groups * per_group * 2 * pitch)
    arr = pin.array.reshape(n, n_groups, per_group, 2, h, w)
    for g in range(n_groups):
        for j in range(per_group):
            q = seqs[(g * per_group + j) % nd]
            arr[:, g, j, 0] = q["left"][:n]
            arr[:, g, j, 1] = q["right"][:n]
    trks = []
    try:
        for _ in range(n_groups):
            trks.append(LockstepTracker(per_group, seqs[0]["K"], seqs[0]["bf"], w, h, max_steps=n, device=local_rank))
        lists = [[([arr[i, g, j, 0] for j in range(per_group)], [arr[i, g, j, 1] for j in range(per_group)]) for g in range(n_groups)] for i in range(n)]
        warm = 2
        for i in range(warm):
            for g, t in enumerate(trks):
                t.step(*lists[i][g])
        for t in trks:
            t.sync()
        barrier()
        t0 = time.perf_counter()
        for i in range(warm, n):
            for g, t in enumerate(trks):
                t.step(*lists[i][g])
        for t in trks:
            t.sync()
        dt = time.perf_counter() - t0          # before the guarded barrier: its handshake is a host round trip
        barrier()
        dt = guard.max(dt)
        untracked = sum(int((t.fetch()[1]["tracked"] == 0).sum()) for t in trks)
    finally:
        for t in trks:
            t.close()
        pin.close()
    return {"workload": "the headline loop with the images in pinned host memory: %d sequences per GPU x %d timed frames, one ps_tracker_step per group and frame "
                        "(every image crosses PCIe inside the timed region)" % (per_group * n_groups, n - warm),
            "tracked_frames_per_s": world * per_group * n_groups * (n - warm) / dt, "ms_per_step": dt / (n - warm) * 1e3, "untracked_frames": untracked,
            "host_to_device_GBps_per_gpu": per_group * n_groups * (n - warm) * 2 * pitch / dt / 1e9}


def _config5_sequence(rank, n_frames):
    """the config-1 generator (two moving boxes, MOTS ids, KITTI-format labels), seed = rank, with the SLOT.MODE 4
    inputs of every frame (instance-id mask, offline detections)"""
    from pointslot_amd import sequence
    job = ("config5", n_frames, rank)
    q = _load_cached(job)
    if q is None:
        q = sequence.generate(n_frames=n_frames, seed=rank)
        q["masks"] = np.stack([sequence.frame_mask(q, i) for i in range(n_frames)])
        q["dets"] = [sequence.frame_detections(q, i) for i in range(n_frames)]
        path = _seq_cache_path(job)
        if path:
            try:
                import pickle
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tmp = "%s.%d.tmp" % (path, os.getpid())
                with open(tmp, "wb") as f:
                    pickle.dump(q, f, protocol=4)
                os.replace(tmp, path)
            except OSError:
                pass
    return q


def config5_leg(rank, world, local_rank, guard, n_frames=154):
    """BASELINE configs[4] as written: one generated 154-frame SLOT.MODE-4 stereo sequence per GPU (seed = rank) through
    the WHOLE per-frame chain - camera chain + object chain, ps_tracker_step_slot_device with one sequence - with a single frame in flight
    (the latency of the chain, not its throughput), then ONE gather of the [154][12] float32 trajectories - the only collective of the
    workflow, timed separately.  The camera chain alone (r05's number for this leg) is timed on the same sequence beside it."""
    import torch
    from pointslot_amd import parallel
    from pointslot_amd.tracker_device import LockstepTracker, pack_detections
    seq = _config5_sequence(rank, n_frames)
    h, w = seq["left"][0].shape
    d = torch.from_numpy(np.stack([seq["left"], seq["right"]], 1)).cuda()
    dm = torch.from_numpy(seq["masks"]).cuda()
    dd = torch.from_numpy(np.stack([pack_detections([seq["dets"][i]], MAX_OBJECTS) for i in range(n_frames)]).view(np.uint8)).cuda()
    res = {}
    for name, objects in (("camera_chain", False), ("slot_chain", True)):
        trk = LockstepTracker(1, seq["K"], seq["bf"], w, h, max_steps=n_frames, device=local_rank, max_objects=MAX_OBJECTS if objects else 0)
        try:
            per = []
            for i in range(n_frames):
                t0 = time.perf_counter()
                if objects:
                    trk.step_slot_device(d[i].data_ptr(), dm[i].data_ptr(), dd[i].data_ptr())
                else:
                    trk.step_device(d[i].data_ptr())
                trk.sync()                                 # one frame in flight: a live sequence delivers its frames one by one
                per.append(time.perf_counter() - t0)
            tcw, st = trk.fetch()
            obj = trk.fetch_objects() if objects else None
        finally:
            trk.close()
        per = np.array(per[1:]) * 1e3                      # the first frame initialises (and pays the first launches)
        res[name] = {"ms_per_frame": float(per.mean()), "median_ms_per_frame": float(np.median(per)), "tcw": tcw, "st": st, "obj": obj}
    tcw, st, obj = res["slot_chain"]["tcw"], res["slot_chain"]["st"], res["slot_chain"]["obj"]
    traj = np.zeros((n_frames, 12), np.float32)
    err = 0.0
    for k in range(n_frames):
        if st["tracked"][k, 0]:
            Rwc = tcw[k, 0, :3, :3].T
            twc = -(Rwc @ tcw[k, 0, :3, 3])
            traj[k] = np.concatenate([Rwc, twc[:, None]], 1).reshape(12)      # System::SaveTrajectoryKITTI row
            err = max(err, float(np.abs(twc - seq["twc"][k][:, 3]).max()))
    guard.barrier()
    t0 = time.perf_counter()
    allt = guard.gather_trajectories(traj)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3
    ms = guard.max(res["slot_chain"]["ms_per_frame"])
    live = obj["id"] >= 0
    return {"workload": "BASELINE configs[4]: %d generated 1242x375 SLOT.MODE-4 stereo sequence(s) x %d frames, one per GPU, camera chain + object chain "
                        "(ps_tracker_step_slot_device, one sequence per handle), one frame in flight per sequence" % (world, n_frames),
            "ms_per_frame": ms, "median_ms_per_frame": guard.max(res["slot_chain"]["median_ms_per_frame"]), "frames_per_s_all_sequences": world * 1e3 / ms,
            "camera_chain_only_ms_per_frame": guard.max(res["camera_chain"]["ms_per_frame"]),
            "tracked": int(st["tracked"].sum()), "detections": int(live.sum()), "detections_with_object": int((obj["tracked"] != 0).sum()),
            "detections_track_ok": int((obj["track_ok"] != 0).sum()),
            "max_abs_position_error_m": guard.max(err), "trajectory_gather_ms": gather_ms,
            "gathered": [list(a.shape) for a in allt]}


class _CAbiTimer:
    """wall time spent inside the C-ABI (every ps_* export the python mirror calls), by entry point: the ctypes function objects of
    `_lib.lib` are replaced by timing wrappers for the life of the context (the mirror looks them up per call)"""

    def __init__(self):
        import re
        from pointslot_amd._lib import lib
        with open(os.path.join(ROOT, "include", "pointslot_hip.h")) as f:
            self.names = sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", f.read())))
        self.lib, self.acc, self.cnt, self.saved = lib, {}, {}, {}

    def __enter__(self):
        for n in self.names:
            try:
                orig = getattr(self.lib, n)
            except AttributeError:
                continue
            self.saved[n] = orig

            def w(*a, _o=orig, _n=n):
                t0 = time.perf_counter()
                r = _o(*a)
                self.acc[_n] = self.acc.get(_n, 0.0) + time.perf_counter() - t0
                self.cnt[_n] = self.cnt.get(_n, 0) + 1
                return r
            setattr(self.lib, n, w)
        return self

    def __exit__(self, *exc):
        for n, o in self.saved.items():
            setattr(self.lib, n, o)

    def reset(self):
        self.acc.clear(); self.cnt.clear()


def drop_in_api_leg(rank, local_rank, n_frames=154, py_frames=40):
    """The drop-in boundary itself (/root/reference/Examples/Stereo/stereo_kitti.cc:108-160 prints exactly this: the
    median / mean tracking time per frame through the classes).  (a) build/stereo_kitti = examples/stereo_kitti.cpp: the reference driver's
    loop on the shim classes - ORB_SLAM2::ORBextractor x 2 on two threads, ComputeStereoMatches, ORBmatcher::SearchByProjection,
    Optimizer::PoseOptimization, one C-ABI call per reference call, HOST images (every frame crosses PCIe, every result comes back) - over the
    config-5 sequence written to disk; it prints where a frame's time goes.  (b) the same frames WITH masks and detections through the
    per-call chain of SLOT.MODE 4 (adds OpencvORBDetector x 2, ComputeObjStereoMatches, SearchByBruceMatching, CFSE3 x 2, object
    SearchByProjection, DynamicStaticDiscrimination): the host side of that chain exists in python only (pointslot_amd/object_tracker.py - the
    tracking harness is frozen), so it reports the wall time INSIDE the C-ABI calls per frame - kernels, PCIe, sync, the
    library's own packing: what any host language pays - beside the python host time, which a C++ caller would not pay."""
    import re
    import shutil
    import subprocess
    import tempfile
    from pointslot_amd import sequence
    seq = _config5_sequence(rank, n_frames)
    h, w = seq["left"][0].shape
    out = {"workload": "the config-5 sequence (%d frames, 1242x375) through the reference-signature classes, host images, one C-ABI call per reference call" % n_frames}
    exe = os.path.join(ROOT, "build", "stereo_kitti")
    tmp = tempfile.mkdtemp(prefix="ps_dropin_")
    try:
        sequence.write_pgm(tmp, seq)
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(local_rank)) if "HIP_VISIBLE_DEVICES" not in os.environ and local_rank else None
        r = subprocess.run([exe, tmp], capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            raise RuntimeError("build/stereo_kitti failed: " + (r.stderr or r.stdout)[-300:])
        med = re.search(r"median tracking time: ([0-9.eE+-]+) ms", r.stdout)
        mean = re.search(r"mean tracking time: ([0-9.eE+-]+) ms", r.stdout)
        sp = re.search(r"split_json: (\{.*\})", r.stdout)
        ok = sum(1 for l in r.stdout.splitlines() if l.startswith("frame ") and ": ok " in l)
        cpp = {"median_ms_per_frame": float(med.group(1)), "mean_ms_per_frame": float(mean.group(1)), "frames_with_pose": ok}
        if sp:
            d = json.loads(sp.group(1))
            cpp["split_ms_per_frame"] = d
            # PCIe volume of a frame at this boundary: two images up, two key / descriptor sets + the stereo result down, then the problems
            kern = d["extract_kernels_ms"] + d["search_kernels_ms"] + d["pose_kernels_ms"]
            cpp["kernels_ms_per_frame_excl_stereo"] = kern
            cpp["calls_minus_kernels_ms"] = d["extract_call_ms"] + d["search_call_ms"] + d["pose_call_ms"] - kern
        out["cpp_shim_camera_chain"] = cpp
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # (b) the SLOT.MODE-4 chain call by call
    from pointslot_amd.tracker import HipBackend, StereoOdometry
    from pointslot_amd import extractor, matcher, optimizer, object_orb  # noqa: F401  (argtypes are set at import: before the wrappers go in)
    n = min(py_frames, n_frames)
    be = HipBackend(device=local_rank)
    try:
        vo = StereoOdometry(be, seq["K"], seq["bf"], w, h)
        per, inside = [], []
        with _CAbiTimer() as tm:
            for k in range(n):
                tm.reset()
                t0 = time.perf_counter()
                vo.track(seq["left"][k], seq["right"][k], seq["masks"][k], seq["dets"][k])
                per.append(time.perf_counter() - t0)
                inside.append((sum(tm.acc.values()), sum(tm.cnt.values()), dict(tm.acc)))
        per = np.array(per[2:]) * 1e3
        ins = np.array([x[0] for x in inside[2:]]) * 1e3
        by = {}
        for _, _, acc in inside[2:]:
            for kname, v in acc.items():
                by[kname] = by.get(kname, 0.0) + v * 1e3 / len(per)
        out["per_call_slot_chain"] = {"frames": int(len(per)), "c_abi_ms_per_frame": float(np.median(ins)), "c_abi_mean_ms_per_frame": float(ins.mean()),
                                      "c_abi_calls_per_frame": float(np.mean([x[1] for x in inside[2:]])),
                                      "python_host_ms_per_frame": float(np.median(per - ins)), "wall_ms_per_frame_python_driver": float(np.median(per)),
                                      "c_abi_ms_per_frame_by_entry_point": {k: round(v, 4) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]},
                                      "tracked": int(sum(1 for t in vo.trajectory if t is not None))}
    finally:
        be.close()
    return out


def fp64_mfma_peak(local_rank):
    """the FP64 denominator is not in the local guide; measured with a v_mfma_f64_16x16x4_f64 microbenchmark."""
    import ctypes
    from pointslot_amd._lib import lib, check
    lib.ps_debug_mfma_f64_peak.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    v = ctypes.c_double(0)
    check(lib.ps_debug_mfma_f64_peak(local_rank, ctypes.byref(v)))
    return v.value


LINE_LIMIT = 8192             # the driver reads the bench line from a bounded tail of stdout: r04's 28.8 KB line came back unparsed


def _num(v, sig=6):
    """floats to `sig` significant digits, containers recursively (the line is a record, not a log)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, dict):
        return {k: _num(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_num(x, sig) for x in v]
    return str(v)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "avg_launch_ms", "images_per_launch")


def compact_line(full):
    """The ONE stdout line of the contract, from the full result: the schema keys, `roofline` + `cpu_baseline`, the whole-step
    roofline, a numeric-only per-stage table, `metric_ba` and the parity spot check - no prose beyond one short `workload` / `sample`
    string each.  Everything else (notes, secondary legs, the per-kernel tables) is in the side file named by `full`."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype"))
    line["data"] = "synthetic"
    cfg = full.get("config", {})
    line["config"] = dict(_pick(cfg, ("sequences_per_gpu", "lockstep_groups_per_gpu", "images_per_step_per_gpu", "object_chain", "dynamic_static_discrimination", "scene")),
                          workload="%s lockstep 1242x375 stereo sequences/GPU, one frame of each per step: ORB 2000 kp x2, stereo, SearchByProjection + PoseOptimization x2%s"
                                   % (cfg.get("sequences_per_gpu"), ", object chain (SLOT.MODE 4)" if cfg.get("object_chain") else ""),
                          parallelism="sequences sharded over %s GPU(s), no data-path collective" % full.get("n_gpus"))
    line["roofline"] = _pick(full.get("roofline") or {}, _ROOF_KEYS)
    if full.get("step_roofline"):
        line["step_roofline"] = _pick(full["step_roofline"], ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_step", "traffic"))
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(_pick(cb, ("value", "unit", "cores", "kind")), sample=str(cb.get("sample", ""))[:100])
    for k in ("cpu_baseline_reference_thread_model", "cpu_baseline_all_cores"):
        if isinstance(full.get(k), dict) and "value" in full[k]:
            line[k] = _pick(full[k], ("value", "unit", "cores"))
    # per stage of the timed step: [stage, bound, ms per step of group 0, HBM fraction or null, HBM traffic / algorithmic bytes or null]
    line["rooflines_columns"] = ["stage", "bound", "ms_per_step", "frac", "traffic_over_algorithmic"]
    line["rooflines"] = [[r.get("stage"), r.get("bound"), r.get("ms_per_step"), r.get("frac"),
                          (r["traffic"] / r["algorithmic_bytes_per_step"]) if r.get("traffic") and r.get("algorithmic_bytes_per_step") else None]
                         for r in full.get("rooflines", [])]
    alone = full.get("roofline_kernels_alone")
    if alone:
        line["kernels_alone"] = {"ms_per_step": alone.get("ms_per_step"),
                                 "frac": {k["kernel"]: k.get("frac") for k in alone.get("kernels", [])},
                                 "stage_ms_per_512_sequences": alone.get("stage_ms_per_512_sequences")}
    tc = full.get("tracking_checks")
    if tc:
        line["tracking_checks"] = {k: v for k, v in tc.items() if k != "checked"}
    mb = full.get("metric_ba")
    if mb:
        line["metric_ba"] = dict(_pick(mb, ("metric", "value", "unit", "higher_is_better")),
                                 roofline=_pick(mb.get("roofline") or {}, ("bound", "achieved", "peak", "unit", "frac", "frac_per_iteration", "traffic")),
                                 cpu_baseline=_pick(mb.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind")))
        sec = full.get("secondary_metrics") or {}
        if "ms_per_iter_1_object" in sec.get("object_ba", {}):
            line["metric_ba"]["value_1_object"] = sec["object_ba"]["ms_per_iter_1_object"]
    if "parity_spot" in full:
        line["parity_spot"] = full["parity_spot"]
    sec = full.get("secondary_metrics") or {}
    # BASELINE configs[4] as written (one SLOT.MODE-4 sequence per GPU, one frame in flight) and the drop-in boundary (the reference driver's loop
    # on the shim classes, host images): ms per frame; the split is wall time inside the C-ABI calls by kind / the kernels' own time / host marshalling
    st = sec.get("sequence_tracking")
    if isinstance(st, dict) and "ms_per_frame" in st:
        line["config5_one_sequence_per_gpu"] = _pick(st, ("ms_per_frame", "median_ms_per_frame", "camera_chain_only_ms_per_frame", "frames_per_s_all_sequences", "tracked",
                                                          "detections_track_ok", "trajectory_gather_ms"))
    di = sec.get("drop_in_api")
    if isinstance(di, dict) and ("cpp_shim_camera_chain" in di or "per_call_slot_chain" in di):
        c, pc = di.get("cpp_shim_camera_chain") or {}, di.get("per_call_slot_chain") or {}
        line["drop_in_api"] = {"cpp_shim_camera_chain": dict(_pick(c, ("median_ms_per_frame", "mean_ms_per_frame", "frames_with_pose")), split_ms=c.get("split_ms_per_frame")),
                               "per_call_slot_chain": _pick(pc, ("c_abi_ms_per_frame", "c_abi_calls_per_frame", "python_host_ms_per_frame", "frames", "tracked"))}
    if sec.get("failed_legs"):
        line["failed_legs"] = sec["failed_legs"]
    line["full"] = full.get("full_result_file")
    line = _num(line, 5)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # never hand the driver a line it cannot read: drop the optional tables, largest first
        for k in ("kernels_alone", "rooflines", "rooflines_columns", "tracking_checks", "cpu_baseline_all_cores", "cpu_baseline_reference_thread_model", "drop_in_api"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    return text


def emit(full):
    """rank 0: the full result to bench_full.json (beside this file; gpurun_out/ too when it exists) and to stderr, then the compact
    line as the LAST line of stdout"""
    full["full_result_file"] = "bench_full.json"
    blob = json.dumps(_num(full, 9), allow_nan=False)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(blob + "\n")
            except OSError:
                pass
    sys.stderr.write(blob + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sequences", type=int, default=1536, help="independent stereo sequences tracked in lockstep per GPU (r05: 768 -> 1536 = three groups of 512: 40.1 -> 41.2 k frames/s on one box; 2304 / 3072: 41.6 / 41.4 k, profiles/r05_group_sweep.txt)")
    ap.add_argument("--groups", type=int, default=3, help="lockstep groups per GPU (one tracker handle and stream each: a group's latency-bound kernels run under the others' issue-bound ones; measured 1 / 2 / 3 / 4 / 6 groups of 256: 30.5 / 37.7 / 40.5 / 37.6 / 40.2 k frames/s, profiles/r04_group_sweep.txt)")
    ap.add_argument("--texture", choices=["kitti", "synthetic"], default="kitti", help="texture of the generated sequences of the headline run")
    ap.add_argument("--scene", choices=["drive", "lateral"], default="drive", help="generator of the headline sequences: forward drive with yaw / lateral translation")
    ap.add_argument("--distinct", type=int, default=32, help="distinct generated sequences per GPU (the tracked ones cycle through them)")
    ap.add_argument("--no-objects", action="store_true", help="headline without the object chain (camera chain only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the headline loop")
    ap.add_argument("--no-alone", action="store_true", help="skip the single-group pass behind the timed region (profiling runs: the kernel trace then holds the timed loop only)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: this process only starts N ranks (fresh children, one per GPU)
    # and forwards rank 0's JSON line; it never touches the GPU itself.  PS_BENCH_LAUNCH_ONLY=1 (CPU test of this path):
    # every rank prints what it was started with and exits before anything imports torch.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from pointslot_amd import parallel
        sys.exit(parallel.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=1800))   # a rank that dies inside a leg must not leave the others waiting for ever
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
    if args.sequences < args.groups or args.sequences % args.groups:
        raise SystemExit("--sequences must be a multiple of --groups")
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
        import datetime
        # (a collective that a dead peer never joins ends after ten minutes instead of the default thirty)
        if share:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=10))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from pointslot_amd import parallel
    # the secondary legs' collectives go through a guard: a leg that raises on ONE rank (before, between or after its collectives) comes
    # back as an error entry on EVERY rank and the next leg runs - nobody is left waiting inside a collective for the launcher's timeout
    guard = parallel.Guard(dist, RED_DEV)

    def gbarrier():
        torch.cuda.synchronize()
        guard.barrier()
        torch.cuda.synchronize()

    with_cpu = not args.no_cpu and rank == 0 and world == 1
    objects = not args.no_objects
    head = tracking_leg(rank, local_rank, args.texture, args.steps, args.warmup, args.sequences, args.groups, barrier, scene=args.scene,
                        n_distinct=args.distinct, objects=objects)
    dt = parallel.max_over_ranks(dist, head["dt"], RED_DEV)
    untracked = int(parallel.max_over_ranks(dist, head["untracked_frames"], RED_DEV))
    err = parallel.max_over_ranks(dist, head["max_abs_position_error_m"], RED_DEV)
    tracked_timed = parallel.sum_over_ranks(dist, head["tracked_frames_timed"], RED_DEV)
    overflowed = int(parallel.sum_over_ranks(dist, head["overflowed_frames"], RED_DEV))

    # the kernels of the step on their own: a short single-group pass over the same sequences (one stream: no overlap between kernels)
    alone = None
    if args.groups > 1 and not args.no_alone:
        # under the guard: a failure here (memory beside the cached sequences, a capacity limit that only shows with every sequence in
        # one group) must not discard the headline that is already measured, nor leave the peers inside a collective
        def alone_pass():
            a1 = tracking_leg(rank, local_rank, args.texture, min(args.steps, 6), max(args.warmup, 2), args.sequences, 1, gbarrier, scene=args.scene,
                              n_distinct=args.distinct, objects=objects, seqs=head["seqs"])
            return {"ms_per_step": guard.max(a1["dt"]) / min(args.steps, 6) * 1e3, "stage_ms": a1["stage_ms_group0"], "images_per_launch": a1["images_per_launch"]}
        alone = guard.run(alone_pass)
        if "error" in alone and len(alone) == 1:
            sys.stderr.write("bench.py: single-group pass failed: %s\n" % alone["error"])
            alone = None
    secondary = None
    if not args.no_secondary:
        secondary = {}
        osteps = min(args.steps, 10)
        sseq = min(args.sequences, 512)   # the secondary tracking legs keep r03's shape: one lockstep group of 512 sequences

        def camera_only():
            o = tracking_leg(rank, local_rank, args.texture, osteps, max(args.warmup, 2), sseq, 1, gbarrier, scene=args.scene,
                             n_distinct=args.distinct, objects=False, seqs=head["seqs"])
            odt = guard.max(o["dt"])
            return {"workload": "the headline loop without masks / detections: the camera chain alone on all keypoints (r02's headline definition); one lockstep group, like every secondary leg "
                                "(two groups on two streams are faster in a fresh process - the headline - and were slower than one, 29 - 33 k against 36.6 k on the lateral scene, behind other "
                                "legs in the same process: stream-to-hardware-queue placement)",
                    "tracked_frames_per_s": world * o["frames_per_step_per_gpu"] * osteps / odt, "ms_per_step": odt / osteps * 1e3,
                    "untracked_frames": o["untracked_frames"], "max_abs_position_error_m": o["max_abs_position_error_m"],
                    "stage_ms_group0": {k: round(v, 5) for k, v in o["stage_ms_group0"].items()}}

        def lateral_scene():
            o = tracking_leg(rank, local_rank, args.texture, osteps, max(args.warmup, 2), sseq, 1, gbarrier, scene="lateral",
                             n_distinct=4, objects=objects)
            odt = guard.max(o["dt"])
            return {"workload": "the headline loop on r02's scene: lateral translation over a ruled surface, two moving boxes, 4 distinct sequences",
                    "tracked_frames_per_s": world * o["frames_per_step_per_gpu"] * osteps / odt, "ms_per_step": odt / osteps * 1e3,
                    "untracked_frames": o["untracked_frames"], "max_abs_position_error_m": o["max_abs_position_error_m"], "objects": o["objects"],
                    "stage_ms_group0": {k: round(v, 5) for k, v in o["stage_ms_group0"].items()}}

        def six_objects():
            o = tracking_leg(rank, local_rank, args.texture, osteps, max(args.warmup, 2), sseq, 1, gbarrier, scene="drive",
                             n_distinct=8, objects=True, n_objects=6)
            odt = guard.max(o["dt"])
            return {"workload": "sensitivity of the headline to the number of objects: the drive scene with SIX objects per sequence (three ahead, three at the "
                                "sides; the headline's scenes carry two), 8 distinct sequences, tracker created for %d detections per frame" % MAX_OBJECTS,
                    "tracked_frames_per_s": world * o["frames_per_step_per_gpu"] * osteps / odt, "ms_per_step": odt / osteps * 1e3,
                    "untracked_frames": o["untracked_frames"], "max_abs_position_error_m": o["max_abs_position_error_m"], "objects": o["objects"],
                    "stage_ms_group0": {k: round(v, 5) for k, v in o["stage_ms_group0"].items()}}

        def twelve_objects():
            o = tracking_leg(rank, local_rank, args.texture, osteps, max(args.warmup, 2), min(sseq, 256), 1, gbarrier, scene="drive",
                             n_distinct=8, objects=True, n_objects=12, max_objects=16)
            odt = guard.max(o["dt"])
            return {"workload": "the drive scene with TWELVE objects per sequence (KITTI tracking frames carry up to ~15 detections), 8 distinct sequences, 256 sequences in one lockstep group, "
                                "tracker created for 16 detections per frame and 16 MapObjects per sequence",
                    "tracked_frames_per_s": world * o["frames_per_step_per_gpu"] * osteps / odt, "ms_per_step": odt / osteps * 1e3,
                    "untracked_frames": o["untracked_frames"], "max_abs_position_error_m": o["max_abs_position_error_m"], "objects": o["objects"],
                    "stage_ms_group0": {k: round(v, 5) for k, v in o["stage_ms_group0"].items()}}

        # a failure of a secondary leg must not take the bench line down, and with several ranks it must not leave the others inside a
        # collective: every rank runs every leg under the guard (parallel.Guard; tests/test_parallel_cpu.py fails one rank of two at
        # every kind of point)
        fp64_peak = fp64_mfma_peak(local_rank)
        failed_legs = []
        for name, fn in (("camera_chain_only", camera_only),
                         ("lateral_scene", lateral_scene),
                         ("six_objects_per_sequence", six_objects),
                         ("twelve_objects_per_sequence", twelve_objects),
                         ("orb_extraction", lambda: orb_leg(rank, local_rank, gbarrier, with_cpu)),
                         ("optimizers", lambda: optimizer_legs(rank, world, local_rank, guard, with_cpu, fp64_peak)),
                         ("lockstep_tracking_host_images", lambda: pcie_leg(rank, world, local_rank, guard, head["seqs"], sseq, 1, gbarrier)),
                         ("sequence_tracking", lambda: config5_leg(rank, world, local_rank, guard)),
                         ("drop_in_api", lambda: (drop_in_api_leg(rank, local_rank) if rank == 0 else {}))):
            r = guard.run(fn)
            if isinstance(r, dict) and "error" in r and len(r) == 1:
                failed_legs.append(name)
                secondary[name] = r
            elif name == "optimizers":
                secondary.update(r)
            else:
                secondary[name] = r
        if failed_legs:
            secondary["failed_legs"] = failed_legs
        if rank == 0:
            for name, fn in (("object_kernels", lambda: object_legs(local_rank, fp64_peak)), ("next_rows", lambda: next_rows_leg(local_rank))):
                try:
                    secondary[name] = fn()
                except Exception as e:   # noqa: BLE001
                    secondary[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        frames_per_step = head["frames_per_step_per_gpu"] * world
        stage = head["stage_ms_group0"]
        nimg = head["images_per_launch"]
        S = head["frames_per_step_per_gpu"]
        # ---- one roofline entry per stage / kernel of the timed step (HIP events on the stream the kernels run on) ----
        rl = []

        G = args.groups
        nimg_step = nimg * G     # the committed counter tables are per STEP (all groups); a launch of group 0 handles 1 / G of it

        def hbm(name, ms, algo, note, kernels=None):
            a = algo / (ms * 1e-3) / 1e9 if ms > 0 else None
            traffic, src = pmc_traffic(kernels or name.split("/")[-1], nimg_step)
            if traffic is not None:
                traffic /= G
            rl.append({"stage": name, "bound": "hbm", "ms_per_step": round(ms, 5), "algorithmic_bytes_per_step": algo, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": a / HBM_PEAK_GBS if a else None, "traffic": traffic, "traffic_source": src, "note": note})

        def other(name, ms, bound, note, kernels=()):
            # no byte figure in SURVEY 8d for these: the roofline is the busiest execution unit of the stage's kernels (unit_busy)
            ub = unit_busy(kernels) if kernels else None
            rl.append({"stage": name, "bound": ("unit-busy:" + ub["unit"]) if ub else bound, "ms_per_step": round(ms, 5), "achieved": ub["frac"] if ub else None,
                       "peak": 1.0 if ub else None, "unit": "busy fraction of the busiest unit (vector ALU / scalar unit / LDS)" if ub else None, "frac": ub["frac"] if ub else None,
                       "traffic": None, "unit_busy": ub["busy"] if ub else None, "unit_busy_source": ub["source"] if ub else None, "character": bound, "note": note})
        for k in ("orb_level_fused", "orb_fast_cells", "orb_describe"):
            if "orb/" + k in stage:
                hbm("orb/" + k, stage["orb/" + k], ALGO_BYTES_PER_IMAGE[k] * nimg, "SURVEY 8d bytes per image x %d images; bound by instruction issue (profiles/)" % nimg)
                ub = unit_busy((k,))
                if ub:
                    rl[-1]["unit_busy"] = ub["busy"]; rl[-1]["unit_busy_source"] = ub["source"]
        if "orb/orb_quadtree" in stage:
            other("orb/orb_quadtree", stage["orb/orb_quadtree"], "latency", "serial list semantics of DistributeOctTree; candidate lists only, no pixel bytes", ("orb_quadtree",))
        other("stereo_match", stage.get("stereo_match", 0.0), "latency", "8f-1 ComputeStereoMatches: row-bucket scan + 11 x 11 SAD slide per left keypoint", ("st_bucket", "st_match", "st_median"))
        other("search_by_projection", stage.get("search_by_projection", 0.0), "latency", "a11 + a12: three windowed searches per frame (th 7, its 2 th retry, the local map)",
              ("pj_project", "pj_gather", "pj_resolve"))
        other("pose_optimization", stage.get("pose_optimization", 0.0), "fp64-valu latency",
              "a14: two PoseOptimization calls per frame, one persistent workgroup per frame (dependent FP64 chains at 1 - 2 waves per SIMD); flop-based figure in secondary_metrics.pose_optimization", ("pose_lm",))
        other("track_glue", stage.get("track_glue", 0.0), "latency", "the host side of Tracking::Track between the kernels, on the device", ("trk_",))
        if objects:
            hbm("object_features", stage.get("object_features", 0.0), 2 * 465750 * nimg + 465750 * S,
                "8f-2 + masks: image + object mask of %d images and %d id masks read once; the work is the part of the pyramid a masked keypoint can reach (tile kernels, latency-bound)" % (nimg, S),
                kernels=("ob_masks", "cvb_plan", "cvb_level0", "cvb_resize", "cvb_detect", "cvb_blur", "cvb_select", "cvb_describe"))
            ub = unit_busy(("ob_masks", "cvb_"))
            if ub:
                rl[-1]["unit_busy"] = ub["busy"]; rl[-1]["unit_busy_source"] = ub["source"]
            other("object_stereo_match", stage.get("object_stereo_match", 0.0), "latency", "8f-1 ComputeObjStereoMatches on the object keys", ("st_bucket", "st_match", "st_median"))
            ob = head["objects"] or {}
            other("object_bruteforce", stage.get("object_bruteforce", 0.0), "int-alu", "a10: one problem per tracked detection (last-frame x current features of the object); pairs / s in secondary_metrics.object_kernels",
                  ("bf_topk", "bf_resolve"))
            other("object_cfse3", stage.get("object_cfse3", 0.0), "fp64-valu latency", "a15: two CFSE3ObjStateOptimization calls per frame (after the brute-force matches, after the local-map search)", ("pose_lm",))
            other("object_search_by_projection", stage.get("object_search_by_projection", 0.0), "latency", "a13: SearchByProjection(F, nOrder, MOPs) per tracked detection", ("pj_gather", "pj_resolve"))
            other("object_glue", stage.get("object_glue", 0.0), "latency", "AssignFeatures, TrackMapObject (RANSAC centroid, box fine tuning, MapObjectInit / ReInit), bookkeeping: %s" % json.dumps(ob), ("ob_begin", "ob_track", "ob_after", "ob_finish", "ob_bf_blocks"))
        # the headline roofline is a single KERNEL's (the stages that are several kernels stay in `rooflines`)
        # (by the kernels' OWN times - the single-group pass the roofline is measured in - where that pass ran: in the timed region the
        # event-to-event time of a stage includes the other groups' kernels, and `orb_level_fused` (8 launches) and `orb_fast_cells`
        # (1) are close enough there to change places from run to run)
        def _dom_key(r):
            k = r["stage"]
            if alone is not None and k in alone.get("stage_ms", {}):
                return alone["stage_ms"][k]
            return r["ms_per_step"]
        dom = max((r for r in rl if r["bound"] == "hbm" and r["stage"].startswith("orb/")), key=_dom_key)
        value = tracked_timed / dt
        roofline_timed = {"bound": "hbm", "kernel": dom["stage"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                          "traffic": dom["traffic"], "traffic_source": dom.get("traffic_source"), "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_step"],
                          "avg_launch_ms": dom["ms_per_step"], "images_per_launch": nimg, "issue": pmc_valu_issue(dom["stage"].split("/")[-1], nimg_step, dom["ms_per_step"] * G),
                          "note": "HIP events on group 0's stream over the TIMED region (orb_level_fused runs once per pyramid level: bytes, time and traffic are its 8 launches = %d images). With %d lockstep "
                                  "groups on %d streams the other group's kernels run beside it, so the event-to-event time is not the kernel's own duration: the roofline of the kernel is `roofline`" % (nimg, G, G)}
        if alone is None:
            roofline_main = dict(roofline_timed, note="the single kernel with the largest time per step among those SURVEY 8d prices in bytes (orb_level_fused runs once per pyramid level: bytes, time and "
                                                      "traffic are per step = its 8 launches); every stage is in `rooflines`")
        else:
            kdom = dom["stage"].split("/")[-1]
            ams = alone["stage_ms"]["orb/" + kdom]
            abytes = ALGO_BYTES_PER_IMAGE[kdom] * alone["images_per_launch"]
            atr, asrc = pmc_traffic(kdom, alone["images_per_launch"])
            roofline_main = {"bound": "hbm", "kernel": "orb/" + kdom, "achieved": abytes / (ams * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": abytes / (ams * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": atr, "traffic_source": asrc, "algorithmic_bytes_per_launch": abytes,
                             "avg_launch_ms": ams, "images_per_launch": alone["images_per_launch"], "issue": pmc_valu_issue(kdom, alone["images_per_launch"], ams),
                             "note": "the single kernel with the largest time per step among those SURVEY 8d prices in bytes (orb_level_fused runs once per pyramid level: bytes, time and traffic are per "
                                     "step = its 8 launches), measured live in this run with HIP events on the kernel's stream in a SINGLE-GROUP pass of the same step over the same sequences right after "
                                     "the timed region (%d steps, %.3f ms per step): the timed region runs %d lockstep groups on %d streams, where a kernel's event-to-event time includes the other group's "
                                     "kernels (`roofline_timed_region_group0`); profiles/%s_bench_kernel_stats_1group_timed.csv is the rocprofv3 table of this pass's command (--groups 1)"
                                     % (min(args.steps, 6), alone["ms_per_step"], G, G, PROFILE_ROUND)}
        # the step as a whole against HBM (frames/s "as fraction of HBM roofline"): SURVEY's 8.68 MB per image x the images of
        # a step (+ image, object mask and id mask read once for the object features) over the wall time of a step on this GPU
        step_algo = sum(ALGO_BYTES_PER_IMAGE.values()) * 2 * S + ((2 * 465750 * 2 * S + 465750 * S) if objects else 0)
        step_ms = dt / args.steps * 1e3
        step_tr = _profile_json(PROFILE_ROUND + "_traffic.json")
        step_traffic = None
        if step_tr and step_tr.get("images_per_launch") == nimg_step:
            step_traffic = sum(v.get("hbm_bytes_per_step", 0.0) for v in step_tr.get("kernels", {}).values()) or None
        step_rl = {"bound": "hbm", "achieved": step_algo / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": step_algo / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": step_algo, "traffic": step_traffic,
                   "note": "whole step of one GPU: SURVEY 8d bytes of every image of the step (+ the object-feature inputs) / ms_per_step; traffic = all kernels of the committed counter pass"}
        out = {
            "metric": "tracked frames/sec KITTI stereo 1242x375",
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
            "data": "synthetic (generated stereo sequences with exact ray-cast geometry: %s; texture = %s)" % (
                "forward drive with yaw through a corridor of planes, two moving objects with instance masks and KITTI labels" if args.scene == "drive"
                else "lateral translation over a ruled surface, two moving boxes",
                "the repository's one real KITTI frame, tests/golden/kitti_000212_gray.png" if args.texture == "kitti" else "seeded value noise + rectangles"),
            "config": {"workload": "BASELINE metric 'tracked frames/sec KITTI stereo 1242x375' in SLOT.MODE 4: %d independent 1242x375 stereo sequences per GPU tracked in lockstep (%d distinct "
                                   "generated drives), one stereo frame of each per step through the whole of Tracking::Track's per-frame work on the device - camera chain (BASELINE configs[1] per "
                                   "image: 8-level ORB, 2000 keypoints + rBRIEF; ComputeStereoMatches; configs[2] per frame: SearchByProjection + PoseOptimization twice)%s - images, instance masks "
                                   "and detections resident in HBM; value counts the frames that came out tracked"
                                   % (S, head["n_distinct"], " and object chain (cv::ORB object features, object stereo, SearchByBruceMatching, CFSE3 x 2, object SearchByProjection, DynamicStaticDiscrimination)" if objects else ""),
                       "sequences_per_gpu": S, "lockstep_groups_per_gpu": args.groups, "images_per_step_per_gpu": 2 * S, "object_chain": objects, "dynamic_static_discrimination": objects, "scene": args.scene,
                       "parallelism": "sequences sharded over %d GPU(s), no collective in the data path" % world},
            "tracking_checks": {"untracked_frames": untracked, "frames_with_overflowed_search_windows": overflowed, "max_abs_position_error_m": err, "distinct_sequences_per_gpu": head["n_distinct"],
                                "objects": head["objects"],
                                "checked": "every frame of every sequence: tracked flag, position against the generator's ground truth; objects: "
                                           "detections with a MapObject / with mbTrackOK, cuboid centres of the distinct sequences against the labels"},
            "roofline": roofline_main,
            "step_roofline": step_rl,
            "roofline_timed_region_group0": roofline_timed,
            "rooflines": rl,
            "roofline_kernels_alone": None if alone is None else {
                "what": "the same step with ONE lockstep group of %d sequences on one stream (%d images per launch, kernels never overlap): %.3f ms per step = %.0f frames/s; the "
                        "headline runs %d groups of %d on %d streams - one group's latency-bound kernels (quadtree, pose_lm, glue) under the other's issue-bound ones"
                        % (S, alone["images_per_launch"], alone["ms_per_step"], S / (alone["ms_per_step"] * 1e-3), G, S // G, G),
                "ms_per_step": alone["ms_per_step"], "stage_ms": {k: round(v, 5) for k, v in alone["stage_ms"].items()},
                "stage_ms_per_512_sequences": {k: round(v * 512.0 / S, 5) for k, v in alone["stage_ms"].items()},
                "stage_ms_per_512_sequences_note": "stage_ms x 512 / %d: the stage times of earlier rounds' bench lines were taken on one lockstep group of 512 sequences "
                                                   "(r03: orb_extract 6.7, object_features 3.5, pose_optimization 1.88 ms).  Linear scaling is right for the throughput-bound "
                                                   "stages; pose_lm runs two workgroups per CU (768 problems: two rounds, 512: one), so its true 512-sequence time is "
                                                   "secondary_metrics.lateral_scene.stage_ms_group0 (one group of 512, another scene)" % S,
                "kernels": [{"kernel": k, "bound": "hbm", "avg_ms_per_step": alone["stage_ms"]["orb/" + k],
                             "achieved": ALGO_BYTES_PER_IMAGE[k] * alone["images_per_launch"] / (alone["stage_ms"]["orb/" + k] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ALGO_BYTES_PER_IMAGE[k] * alone["images_per_launch"] / (alone["stage_ms"]["orb/" + k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": pmc_traffic(k, alone["images_per_launch"])[0], "traffic_source": pmc_traffic(k, alone["images_per_launch"])[1]}
                            for k in ("orb_level_fused", "orb_fast_cells", "orb_describe") if "orb/" + k in alone["stage_ms"]] +
                           # the kernels SURVEY 8d does not price in bytes: the busiest unit's busy fraction (committed counter pass), the stage's own time in this pass
                           [dict(kernel=kn, bound="unit-busy:" + ub["unit"], avg_ms_per_step=alone["stage_ms"].get(stg), achieved=ub["frac"], peak=1.0, unit="busy fraction",
                                 frac=ub["frac"], unit_busy=ub["busy"], source=ub["source"])
                            for kn, stg, ub in ((kn, stg, unit_busy((kn,))) for kn, stg in (("pose_lm", "pose_optimization"), ("orb_quadtree", "orb/orb_quadtree"), ("st_match", "stereo_match"),
                                                                                             ("pj_gather", "search_by_projection"), ("pj_resolve", "search_by_projection"),
                                                                                             ("cvb_plan", "object_features"), ("cvb_resize", "object_features"), ("cvb_detect", "object_features"),
                                                                                             ("cvb_blur", "object_features"), ("cvb_select", "object_features"), ("cvb_describe", "object_features"),
                                                                                             ("ob_masks", "object_features"), ("bf_topk", "object_bruteforce"), ("bf_resolve", "object_bruteforce"),
                                                                                             ("trk_begin", "track_glue"), ("ob_track", "object_glue"))) if ub]},
            "stage_ms": {k: round(v, 5) for k, v in stage.items()},
            "stage_ms_note": "HIP events on group 0's stream over the timed steps; with %d groups the stages of different groups overlap in time" % args.groups,
        }
        if with_cpu:
            one, threaded, allc, worst, checked, oc, ob_bad = cpu_tracking_baseline(head["seqs"], head["tcw_group0"], head.get("obj_group0"), objects)
            out["cpu_baseline"] = one
            out["cpu_baseline_reference_thread_model"] = threaded
            out["cpu_baseline_all_cores"] = allc
            out["parity_spot"] = "green" if (checked > 0 and worst < 1e-4 and ob_bad == 0) else "red"
            out["parity_spot_detail"] = ("%d frames of the timed run: Tcw vs the CPU restatement of the same loop, max |diff| %.3g (bar 1e-4: float32 poses of an FP64 LM chained over the "
                                         "frames); %d object records (features, MapObject, brute-force / projection match counts, inliers) compared, %d differ" % (checked, worst, oc, ob_bad))
        if secondary is not None:
            out["secondary_metrics"] = secondary
            if "ms_per_iter" in secondary.get("object_ba", {}):
                ba = secondary["object_ba"]
                out["metric_ba"] = {"metric": "ms/iter 50-KF object BA (8 objects)", "value": ba["ms_per_iter"], "unit": "ms", "higher_is_better": False,
                                    "roofline": ba.get("roofline"),
                                    "cpu_baseline": {"value": ba.get("cpu_port_ms_per_iter_1core_1object"), "unit": "ms/iter (one object)", "cores": 1, "kind": "port",
                                                     "sample": "one object's 5 + 10 iteration schedule through the CPU restatement (oracle/opt_oracle.cpp)"}}
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
