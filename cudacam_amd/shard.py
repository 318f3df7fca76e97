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


def gather_in_order(local_maps, n_frames, dist=None, dst=0):
    """Reassembles per-rank results on rank `dst`, in frame order.  `local_maps`: this rank's array of shape
    (stop - start, ...) for its frame_range block.  Returns the (n_frames, ...) array on `dst`, None elsewhere.
    Frames are independent, so this gather is the only step in which ranks exchange image data at all -- and a stream
    consumer that reads each rank's maps where they are does not need it (bench.py does not)."""
    import numpy as np
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local_maps)
    world, rank = dist.get_world_size(), dist.get_rank()
    start, stop = frame_range(n_frames, rank, world)
    if len(local_maps) != stop - start:
        raise ValueError(f"rank {rank} holds {len(local_maps)} maps for frames [{start}, {stop})")
    parts = [None] * world if rank == dst else None
    dist.gather_object((start, stop, np.asarray(local_maps)), parts, dst=dst)
    if rank != dst:
        return None
    parts.sort(key=lambda p: p[0])
    expect = 0
    for a, b, maps in parts:   # contiguous, complete, no overlap
        if a != expect or len(maps) != b - a:
            raise ValueError("ranks do not tile the frame stream")
        expect = b
    if expect != n_frames:
        raise ValueError("ranks do not cover the frame stream")
    return np.concatenate([p[2] for p in parts if len(p[2])], axis=0)


def run_on_devices(frames, devices, low=10, high=40, mode=0, batch=None, options=()):
    """In-process counterpart of the one-rank-per-GPU launch: `frames` (n, H, W[, 3]) u8 are cut into contiguous blocks,
    one per entry of `devices` (HIP device ordinals; an ordinal may appear more than once), each block goes through its
    own context -- own stream, own device buffers -- on its own host thread, and the edge maps come back in frame order.
    No device ever sees another device's frames (independent units, no collective)."""
    import threading

    import numpy as np

    from . import api
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape[:3]
    ch = 1 if frames.ndim == 3 else frames.shape[3]
    world = len(devices)
    # per-channel mode (BASELINE configs[4]) returns three edge maps per input frame: frame f's maps are out[3f .. 3f+2]
    per_frame = 3 if any(opt == api.OPT_PER_CHANNEL and val for opt, val in options) else 1
    if per_frame == 3 and ch != 3:
        raise ValueError("OPT_PER_CHANNEL needs 3-channel frames")
    out = np.empty((n * per_frame, h, w), np.uint8)
    errors = []

    def work(rank):
        try:
            a, b = frame_range(n, rank, world)
            if a == b:
                return
            step = min(batch or (b - a), b - a)
            with api.Context(w, h, ch, step, mode, device=devices[rank]) as ctx:
                ctx.set_thresholds(low, high)
                for opt, val in options:
                    ctx.set_option(opt, val)
                for s in range(a, b, step):
                    e = min(b, s + step)
                    out[s * per_frame:e * per_frame] = ctx.process(frames[s:e])
        except Exception as exc:   # surfaced on the calling thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out
