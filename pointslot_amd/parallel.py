"""Multi-GPU plumbing.  The hot path shards with NO exchange inside the computation:
whole sequences, frames of an offline batch and per-object BA problems are independent units, so every rank
(one process per GPU) works on its own slice; the only collective is the gather of the results at the end
(trajectories [frames][12] fp32, a few tens of KB — one RCCL all_gather over xGMI, latency-bound) and the
max-over-ranks of the timing.  Works with backend "nccl" (= RCCL on ROCm) and with "gloo" (CPU tests)."""
import os

import numpy as np


def shard_units(n_units, world, rank):
    """Contiguous, balanced partition of units 0..n_units-1 over `world` ranks: the first n_units % world ranks
    get one extra unit.  Returns range(begin, end) for `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank %r/%r" % (world, rank))
    base, extra = divmod(n_units, world)
    begin = rank * base + min(rank, extra)
    return range(begin, begin + base + (1 if rank < extra else 0))


def round_robin_units(n_units, world, rank):
    """Round-robin assignment (sequence i -> GPU i % world), the throughput-run layout of SURVEY.md 8e."""
    return range(rank, n_units, world)


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1.  Returns (dist or None,
    rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return None, rank, world, local_rank
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return dist, rank, world, local_rank


def max_over_ranks(dist, value, device="cpu"):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_trajectories(dist, local, device="cpu"):
    """local: float32 array [n_local_frames, 12] (rows of Tcw as System::SaveTrajectoryKITTI writes them,
    /root/reference/src/System.cc:400-402).  Ranks may hold different frame counts.  Returns the list of per-rank
    arrays on every rank (all_gather of padded buffers + counts)."""
    import torch
    local = np.ascontiguousarray(local, np.float32).reshape(-1, 12)
    if dist is None:
        return [local]
    world = dist.get_world_size()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = max(int(c.item()) for c in counts)
    buf = torch.zeros((max(nmax, 1), 12), dtype=torch.float32, device=device)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(device)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [o[:int(c.item())].cpu().numpy() for o, c in zip(outs, counts)]


class PeerFailed(BaseException):   # not an Exception: a broad `except Exception` inside a leg must not swallow it (the handshakes would fall out of step)
    """another rank reported a failure at the handshake in front of a collective"""


class Guard:
    """Collectives of a leg that may fail on ONE rank without leaving the others waiting inside a collective.

    Every collective issued through the guard is preceded by a handshake (an all-reduce MIN of an ok flag).  A rank whose leg
    raises does exactly one handshake with ok = 0 (at the end of `run`); the others meet it at their next handshake - the one in
    front of their next collective, or the one at the end of their own `run` - learn of the failure there, abandon the leg without
    a further handshake, and every rank carries on with the next leg: the handshake counts of all ranks stay in step whatever the
    point of failure.  With dist = None (one rank) it only catches the exception."""

    def __init__(self, dist, device="cpu"):
        self.dist, self.device = dist, device

    def _handshake(self, ok):
        if self.dist is None:
            return bool(ok)
        import torch
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def check(self):
        if not self._handshake(True):
            raise PeerFailed("a peer rank failed in this leg")

    def max(self, value):
        self.check()
        return max_over_ranks(self.dist, value, self.device)

    def sum(self, value):
        self.check()
        return sum_over_ranks(self.dist, value, self.device)

    def gather_trajectories(self, local):
        self.check()
        return gather_trajectories(self.dist, local, self.device)

    def barrier(self):
        self.check()
        if self.dist is not None:
            self.dist.barrier()

    def run(self, fn):
        """fn() -> result, or {"error": ...} on every rank when any rank failed inside it"""
        try:
            r = fn()
        except PeerFailed as e:
            return {"error": "PeerFailed: %s" % e}
        except Exception as e:   # noqa: BLE001
            self._handshake(False)
            return {"error": "%s: %s" % (type(e).__name__, e)}
        if not self._handshake(True):
            return {"error": "PeerFailed: a peer rank failed in this leg"}
        return r


def launch_ranks(script, argv, n_ranks, timeout=None, extra_env=None):
    """`python bench.py --gpus N` without a launcher: starts N fresh child processes of `script` (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, the layout `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N` would give them), forwards rank 0's stdout and returns the largest exit code.  The calling process
    must not have initialised the GPU (it only waits); every child is killed if one of them fails or the timeout expires."""
    import socket
    import subprocess
    import sys
    import time
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.update(extra_env or {})
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        out = None if r == 0 else subprocess.DEVNULL       # rank 0 prints the JSON line; stderr of every rank is inherited
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    if code != 0:
                        rc = max(rc, code if code > 0 else 1)
            if rc != 0 or (deadline is not None and time.time() > deadline):
                if pending and rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return rc
