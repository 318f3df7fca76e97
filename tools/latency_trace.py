#!/usr/bin/env python3
"""One-frame-per-call loop (the reference's pattern, cvPipeline.cpp:19-41) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/latency_trace.py [frames_per_call]
then tools/trace_summary.py <dir> shows the kernels of one call.  Prints the host-side ms per call."""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cudacam_amd import api, synth  # noqa: E402

W, H = 1920, 1080
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pipeline = int(sys.argv[2]) if len(sys.argv) > 2 else 0
api.preload_hip_runtime()
import numpy as np  # noqa: E402
import torch  # noqa: E402

frames = np.stack([synth.natural(W, H, 5 + f) for f in range(min(nb, 4))])
d_in = torch.from_numpy(np.tile(frames, ((nb + 3) // 4, 1, 1))[:nb].copy()).cuda()
d_outs = [torch.zeros_like(d_in) for _ in range(4)]
torch.cuda.synchronize()
with api.Context(W, H, 1, nb) as ctx:
    ctx.set_option(api.OPT_PIPELINE, pipeline)
    if len(sys.argv) > 3:
        ctx.set_option(api.OPT_FRONT_WPB, int(sys.argv[3]))   # 1 / 4: waves per workgroup of k_front8 (pipelined runs)
    for k in range(30):
        ctx.run_device(d_in.data_ptr(), W, W * H, d_outs[k % 4].data_ptr(), W, W * H, nb)
        if not pipeline:
            ctx.sync()
    ctx.sync()
    N = 200
    t0 = time.perf_counter()
    for k in range(N):
        ctx.run_device(d_in.data_ptr(), W, W * H, d_outs[k % 4].data_ptr(), W, W * H, nb)
        if not pipeline:
            ctx.sync()
    ctx.sync()
    dt = (time.perf_counter() - t0) / N
    print(f"{nb} frame(s) per call, pipeline {pipeline}: {dt * 1e3:.4f} ms per call, {nb / dt:.0f} frames/s; hysteresis {ctx.hysteresis_info()}")
