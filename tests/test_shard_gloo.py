"""N>1 path on CPU: world_size-2 (and 3) gloo processes shard a stream of REAL frames exactly as bench.py's ranks do
(cudacam_amd/shard.py), each rank runs its block through the per-frame function -- on CPU that is the oracle, the
product has no CPU path -- rank 0 reassembles the maps in frame order (shard.gather_in_order) and compares them with
an unsharded run of the same function; plus the barrier and the MAX-reduced timing bench.py reports against."""
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


def _frame_worker(rank, world, port, n_frames, out_q):
    """One bench.py rank in miniature: own block of the frame stream, no data-path collective, results gathered in order."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cudacam_amd import synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert dist.get_world_size() == world
    a, b = shard.frame_range(n_frames, rank, world)
    frames = [synth.natural(96, 64, 100 + f) if f % 3 else synth.noise(96, 64, 100 + f) for f in range(a, b)]   # every rank can make any frame: no scatter
    maps = np.stack([O.canny_r(f, 10, 40) for f in frames]) if frames else np.zeros((0, 64, 96), np.uint8)
    dist.barrier()
    whole = shard.gather_in_order(maps, n_frames, dist, dst=0)
    t = shard.reduce_max_seconds(0.125 * (rank + 1), dist)
    if rank == 0:
        out_q.put((whole, t))
    else:
        assert whole is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 5), (2, 1)])
def test_sharded_frames_reassemble_in_order_gloo(oracle, world, n_frames):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frame_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    whole, t = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from cudacam_amd import synth
    want = np.stack([oracle.canny_r(synth.natural(96, 64, 100 + f) if f % 3 else synth.noise(96, 64, 100 + f), 10, 40) for f in range(n_frames)])
    assert whole.shape == want.shape and np.array_equal(whole, want)   # byte-identical to the unsharded run, in frame order
    assert abs(t - 0.125 * world) < 1e-9                                # MAX over ranks


def test_gather_in_order_rejects_a_wrong_block():
    with pytest.raises(ValueError):
        class _D:   # a one-rank "group" that claims two ranks
            @staticmethod
            def is_initialized(): return True
            @staticmethod
            def get_world_size(): return 2
            @staticmethod
            def get_rank(): return 0
        shard.gather_in_order(np.zeros((5, 2, 2), np.uint8), 7, _D)   # rank 0 of 2 owns 4 frames of 7, not 5


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
