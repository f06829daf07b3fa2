"""CPU tests of the multi-GPU plumbing with the gloo backend, world_size 2 (one process per rank)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointslot_amd import parallel  # noqa: E402


def test_shard_units_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(parallel.shard_units(n, world, r))
            assert seen == list(range(n))
            sizes = [len(parallel.shard_units(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
            rr = sorted(i for r in range(world) for i in parallel.round_robin_units(n, world, r))
            assert rr == list(range(n))


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from pointslot_amd import parallel
dist, rank, world, local = parallel.init_from_env("gloo")
assert world == 2 and dist is not None
units = list(parallel.shard_units(5, world, rank))          # 5 sequences over 2 ranks -> 3 + 2
traj = np.stack([np.full(12, 100 * u + k, np.float32) for u in units for k in range(4 + u)]) if units else np.zeros((0, 12), np.float32)
parts = parallel.gather_trajectories(dist, traj)
assert len(parts) == 2
all_rows = np.concatenate(parts)
exp = np.stack([np.full(12, 100 * u + k, np.float32) for u in range(5) for k in range(4 + u)])
assert np.array_equal(all_rows, exp), (all_rows.shape, exp.shape)
t = parallel.max_over_ranks(dist, 1.5 + rank)
assert t == 2.5
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gloo_world2_gather_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


GUARD_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
from pointslot_amd import parallel
dist, rank, world, local = parallel.init_from_env("gloo")
g = parallel.Guard(dist)
mode = sys.argv[1]
def leg():
    a = g.max(1.0 + rank)                        # collective 1
    if mode == "before" and rank == 1: raise ValueError("boom before the first collective")
    if mode == "middle" and rank == 1: raise ValueError("boom between two collectives")
    b = g.sum(2.0)                               # collective 2
    if mode == "end" and rank == 0: raise ValueError("boom after the last collective")
    return {"a": a, "b": b}
def leg_first():
    if mode == "before" and rank == 1: raise ValueError("boom before the first collective")
    return {"a": g.max(1.0 + rank), "b": g.sum(2.0)}
r = g.run(leg_first if mode == "before" else leg)
if mode == "none":
    assert r == {"a": 2.0, "b": 4.0}, r
else:
    assert "error" in r, r                       # EVERY rank learns of the failure ...
after = g.run(lambda: g.max(10.0 + rank))        # ... and the next leg's collectives line up again
assert after == 11.0, after
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", r)
'''


def test_a_leg_that_fails_on_one_rank_fails_on_all_and_the_next_leg_runs(tmp_path):
    """bench.py's secondary legs: a rank-local exception before, between or after the collectives of a leg must neither hang the other
    ranks nor desynchronise the following legs (parallel.Guard)."""
    script = tmp_path / "guard_worker.py"
    script.write_text(GUARD_WORKER % ROOT)
    for k, mode in enumerate(("none", "before", "middle", "end")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + k), WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, str(script), mode], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(2)]
        outs = [p.communicate(timeout=120)[0].decode() for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0, (mode, o)
            assert "rank %d ok" % r in o, (mode, o)


def _bench(args, **env):
    e = dict(os.environ, PS_BENCH_LAUNCH_ONLY="1", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks itself (it silently ran one)."""
    import json
    r = _bench(["--gpus", "3", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # only rank 0's line is forwarded
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["rank"] == 0 and out["master"].startswith("127.0.0.1:")
    # a failing rank fails the launcher
    r = _bench(["--gpus", "2"], PS_BENCH_LAUNCH_FAIL_RANK="1")
    assert r.returncode != 0
    # started by an external launcher with a different world size: loud failure, not a silent single-rank run
    r = _bench(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus is 4" in (r.stderr + r.stdout)
    r = _bench(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0
