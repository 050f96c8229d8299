"""world_size-2 gloo tests (CPU) of the N>1 paths: (1) window sharding + the bench timing
aggregation, (2) landmark sharding of one window: the sum all-reduce of the ranks' partial reduced
systems equals the unsharded reduced system (compute = the CPU oracle; the collective and the
partition logic are what is under test)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, L):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _gfbe_import import gf
    import oracle_lib
    synth, gd = gf.synth, gf.dist
    # (1) window sharding: disjoint cover + aggregation contract
    mine = gd.shard_indices(11, rank, world)
    units, tmax = gd.aggregate_throughput(len(mine), 0.5 + rank, dist)
    owned = [None] * world
    dist.all_gather_object(owned, mine)
    # (2) landmark sharding of one window
    orc = oracle_lib.load()
    snap = synth.Scenario(seed=99, n_landmarks=L, use_wheel=True).window(0)
    full = gd.reduced_system(orc.linearize(snap))
    part = gd.reduced_system(orc.linearize(gd.shard_landmarks(snap, rank, world)))
    t = torch.from_numpy(part.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    err = float(np.abs(t.numpy() - full).max() / np.abs(full).max())
    q.put((rank, units, tmax, owned, err))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, L=120):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, L)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, units, tmax, owned, err in res:
        assert units == 11 and tmax == world - 0.5               # SUM of units, MAX of elapsed
        flat = sorted(i for o in owned for i in o)
        assert flat == list(range(11))                           # disjoint cover
        assert err < 1e-12, err                                  # all-reduced partials == unsharded system


def test_window_and_landmark_sharding_world2():
    _run_world(2)


def test_window_and_landmark_sharding_world3_ragged():
    _run_world(3, L=100)                                         # 34 + 33 + 33 landmarks


def test_landmark_sharding_world3_with_an_empty_shard():
    # two landmarks over three ranks: rank 2 owns none and (not being rank 0) no dense factor either — an all-zero partial;
    # the sum must still be the unsharded system
    _run_world(3, L=2)
