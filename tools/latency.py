#!/usr/bin/env python3
"""Single-frame latency (the reference's own call pattern: one 1080p frame per CannyEdge::run).
Prints host-measured ms per frame for (a) device-resident run + sync, (b) upload + run + download from pageable
memory (what cvp::cvPipeline::process does), and the per-stage device times of the last frame."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cudacam_amd import api, synth  # noqa: E402

W, H = 1920, 1080
img = synth.natural(W, H, 5)
api.preload_hip_runtime()
import torch  # noqa: E402

d_in = torch.from_numpy(img).cuda()
d_out = torch.zeros_like(d_in)
torch.cuda.synchronize()
N = 300
for pipeline in (0,):
    with api.Context(W, H, 1, 1) as ctx:
        ctx.set_option(api.OPT_PIPELINE, pipeline)
        for _ in range(20):
            ctx.run_device(d_in.data_ptr(), W, W * H, d_out.data_ptr(), W, W * H, 1)
            ctx.sync()
        t0 = time.perf_counter()
        for _ in range(N):
            ctx.run_device(d_in.data_ptr(), W, W * H, d_out.data_ptr(), W, W * H, 1)
            ctx.sync()
        dt = (time.perf_counter() - t0) / N
        ctx.enable_profiling(True)
        ctx.run_device(d_in.data_ptr(), W, W * H, d_out.data_ptr(), W, W * H, 1)
        ctx.sync()
        st = [round(ctx.stage_time_ms(s), 4) for s in range(6)]
        print(f"device-resident, run + sync per frame: {dt * 1e3:.4f} ms ({1 / dt:.0f} frames/s); stage ms {st}; hysteresis launches with work {ctx.hysteresis_info()}")
        ctx.enable_profiling(False)
        for _ in range(10):
            ctx.process(img)
        t0 = time.perf_counter()
        for _ in range(N):
            ctx.process(img)
        dt = (time.perf_counter() - t0) / N
        print(f"pageable host frame, upload + run + download: {dt * 1e3:.4f} ms ({1 / dt:.0f} frames/s)")
m = api.cvPipeline(0, W, H, 1)
for _ in range(10):
    m.process(img, api.CannyStage.HYSTER)
t0 = time.perf_counter()
for _ in range(N):
    m.process(img, api.CannyStage.HYSTER)
dt = (time.perf_counter() - t0) / N
print(f"python cvPipeline.process (profiling on, as the reference): {dt * 1e3:.4f} ms")

# the same through page-locked staging buffers (what a pinned-buffer CannyEdge::run would do): copy in, DMA, run, DMA, copy out
import ctypes as C
lib = api.load_library()
with api.Context(W, H, 1, 1) as ctx:
    hin, hout = lib.hc_host_alloc(W * H), lib.hc_host_alloc(W * H)
    out = np.empty((H, W), np.uint8)
    def once():
        C.memmove(hin, img.ctypes.data, W * H)
        api._ck(lib.hc_upload(ctx.handle, C.c_void_p(hin), W, W * H, 1))
        ctx.run(api.CannyStage.HYSTER, 1)
        api._ck(lib.hc_download(ctx.handle, C.c_void_p(hout), W, W * H, 1))
        C.memmove(out.ctypes.data, hout, W * H)
    for _ in range(10):
        once()
    t0 = time.perf_counter()
    for _ in range(N):
        once()
    dt = (time.perf_counter() - t0) / N
    print(f"page-locked staging (memcpy in, DMA, run, DMA, memcpy out): {dt * 1e3:.4f} ms ({1 / dt:.0f} frames/s)")
