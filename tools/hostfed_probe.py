#!/usr/bin/env python3
"""Where does the host-fed pipeline lose the link's duplex rate?  Times, on three contexts in a ring: uploads only,
downloads only, both without the detector in between, and the full loop -- with and without HC_OPT_COPY_STREAMS."""
import ctypes as C
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cudacam_amd import api  # noqa: E402

W, H, NB, D, REPS = 1920, 1080, 32, 3, 45
lib = api.load_library()
for copy_streams in (0, 1):
    ring = []
    for _ in range(D):
        ctx = api.Context(W, H, 1, NB)
        ctx.set_option(api.OPT_COPY_STREAMS, copy_streams)
        hin, hout = lib.hc_host_alloc(W * H * NB), lib.hc_host_alloc(W * H * NB)
        C.memset(hin, 37, W * H * NB)
        api._ck(lib.hc_upload(ctx.handle, C.c_void_p(hin), W, W * H, NB))
        ctx.run(api.CannyStage.HYSTER, NB)
        ctx.sync()
        ring.append((ctx, hin, hout))

    def loop(up, run, down):
        busy = [False] * D
        t0 = time.perf_counter()
        for i in range(REPS):
            k = i % D
            ctx, hin, hout = ring[k]
            if busy[k]:
                if down:
                    api._ck(lib.hc_download_end(ctx.handle))
                else:
                    ctx.sync()
            if up:
                api._ck(lib.hc_upload(ctx.handle, C.c_void_p(hin), W, W * H, NB))
            if run:
                ctx.run(api.CannyStage.HYSTER, NB)
            if down:
                api._ck(lib.hc_download_begin(ctx.handle, C.c_void_p(hout), W, W * H, NB))
            busy[k] = True
        for k in range(D):
            if busy[k]:
                if down:
                    api._ck(lib.hc_download_end(ring[k][0].handle))
                else:
                    ring[k][0].sync()
        dt = time.perf_counter() - t0
        return REPS * NB * W * H / dt / 1e9

    for name, cfg in (("uploads only", (1, 0, 0)), ("downloads only", (0, 0, 1)), ("uploads + downloads, no detector", (1, 0, 1)), ("full loop", (1, 1, 1))):
        loop(*cfg)
        print(f"copy_streams {copy_streams}  {name:34s} {loop(*cfg):6.1f} GB/s each way", flush=True)
    for ctx, hin, hout in ring:
        ctx.close()
        lib.hc_host_free(C.c_void_p(hin))
        lib.hc_host_free(C.c_void_p(hout))
