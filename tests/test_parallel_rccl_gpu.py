"""The multi-GPU plumbing's collectives through RCCL itself (backend "nccl" on ROCm) - with a world of ONE rank, which is all a 1-GPU box
allows: process-group set-up on the device, all_reduce MAX / SUM / MIN and the padded all_gather of parallel.gather_trajectories on device
tensors, the guard's handshake in front of each.  It does not measure anything and says nothing about xGMI: it shows that the calls bench.py
issues for N > 1 are accepted by the library on this image (the N > 1 logic is covered by tests/test_parallel_cpu.py on gloo)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import datetime, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from pointslot_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(minutes=5))
assert dist.get_backend() == "nccl"
dev = "cuda:0"
assert parallel.max_over_ranks(dist, 1.5, dev) == 1.5
assert parallel.sum_over_ranks(dist, 2.25, dev) == 2.25
traj = np.arange(7 * 12, dtype=np.float32).reshape(7, 12)
parts = parallel.gather_trajectories(dist, traj, dev)
assert len(parts) == 1 and np.array_equal(parts[0], traj)
parts = parallel.gather_trajectories(dist, np.zeros((0, 12), np.float32), dev)      # a rank without frames
assert len(parts) == 1 and parts[0].shape == (0, 12)
g = parallel.Guard(dist, dev)
assert g.run(lambda: {"a": g.max(3.0), "b": g.sum(4.0)}) == {"a": 3.0, "b": 4.0}
def bad():
    raise ValueError("boom")
r = g.run(bad)
assert "error" in r, r
assert g.run(lambda: g.max(5.0)) == 5.0                # the next leg's collectives line up
g.barrier()
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl world-1 ok")
'''


@pytest.mark.gpu
def test_rccl_accepts_the_plumbings_collectives_world_of_one(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rccl world-1 ok" in r.stdout
