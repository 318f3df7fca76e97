#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary (hc_upload + hc_run + hc_download), for DESIGN.md; never the bench value."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudacam_amd import api, synth
W, H, B = 1920, 1080, 64
frames = synth.frames("natural", W, H, 8)
host = np.ascontiguousarray(np.tile(frames, (B // 8, 1, 1)))
with api.Context(W, H, 1, B) as ctx:
    out = ctx.process(host)          # warm-up (pageable host memory, as a cv::Mat would be)
    t0 = time.perf_counter()
    for _ in range(5):
        out = ctx.process(host)
    dt = (time.perf_counter() - t0) / 5
print(f"host->device->host, {B} x {W}x{H} frames per call: {B / dt:.0f} frames/s ({dt * 1e3:.1f} ms per call, {2 * W * H * B / dt / 1e9:.1f} GB/s over PCIe both ways)")
