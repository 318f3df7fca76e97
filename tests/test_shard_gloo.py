"""N>1 path on CPU: world_size-2 gloo processes shard a frame stream, each runs the (CPU oracle
stand-in for its) shard, and rank 0 checks order-preserving coverage plus the MAX-reduced timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cudacam_amd import shard


def test_frame_range_partition():
    for n in (0, 1, 7, 8, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard.frame_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            for f in range(0, n, max(1, n // 17)):
                r = shard.owner_of(f, n, world)
                assert spans[r][0] <= f < spans[r][1]


def _worker(rank, world, port, n_frames, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard.frame_range(n_frames, rank, world)
    # every rank "processes" its frames: checksum of the frame ids stands in for the edge maps
    local = torch.tensor([sum(range(a, b)), b - a], dtype=torch.int64)
    gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, local)
    dist.barrier()
    t = shard.reduce_max_seconds(0.25 * (rank + 1), dist)
    if rank == 0:
        out_q.put(([g.tolist() for g in gathered], t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    world, n_frames = 2, 37
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(g[1] for g in gathered) == n_frames
    assert sum(g[0] for g in gathered) == sum(range(n_frames))
    assert abs(t - 0.5) < 1e-9  # MAX over ranks
