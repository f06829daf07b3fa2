#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on N MI355X of one node.

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  the ORB extractor (8-level pyramid, FAST cells, quadtree, 7x7 blur, rBRIEF; BASELINE.json configs[1])
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
    "orb_pyramid_level": 465750 + PADDED_PX,          # input read + padded pyramid write
    "orb_fast_cells": PADDED_PX,                      # FAST reads the padded pyramid once
    "orb_quadtree": 0,                                # candidate lists only (not in the pixel budget)
    "orb_blur": PADDED_PX + LEVEL_PX,                 # blur read + blur write
    "orb_describe": LEVEL_PX + NFEAT * (32 + 28),     # gather (upper bound) + outputs
}
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


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
    return {"value": n / dt, "unit": "frames/s", "cores": 2, "kind": "port",
            "sample": "%d stereo pairs of the step's batch, CPU restatement of the reference algorithm "
                      "(oracle/orb_oracle.cpp, -O3 -march=native), left/right on 2 threads; host has %d cores"
                      % (n, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per step per GPU")
    ap.add_argument("--cpu-pairs", type=int, default=48, help="stereo pairs timed on the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the timed work produced keypoints (outputs stay in HBM; fetch one image)
    kps, desc = ex.fetch(0)
    assert len(kps) >= NFEAT // 2 and desc.shape == (len(kps), 32)

    if rank == 0:
        total_pairs = args.pairs * world * args.steps
        value = total_pairs / dt
        dom = max(stage_ms, key=lambda k: stage_ms[k])
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": dom_ms},
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(batch, args.cpu_pairs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
