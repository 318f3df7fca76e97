"""Frame sharding across the GPUs of one node.

Frames are independent units: rank r of `world` owns a contiguous block of the stream, processes it
on its own device/stream through its own hc_ctx, and no data-path collective is needed.  The only
inter-rank traffic is the barrier around the timed region and the MAX-reduce of the elapsed time
(torch.distributed: RCCL on GPUs, gloo in the CPU tests).
"""


def frame_range(n_frames, rank, world):
    """[start, stop) of the frames rank `rank` owns: contiguous, sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def owner_of(frame, n_frames, world):
    """Rank that owns `frame` (inverse of frame_range)."""
    base, extra = divmod(n_frames, world)
    cut = extra * (base + 1)
    if frame < cut:
        return frame // (base + 1)
    return extra + (frame - cut) // base if base else world - 1


def reduce_max_seconds(elapsed, dist=None, device=None):
    """MAX over ranks of a wall-clock interval (what bench.py reports the throughput against)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
