#!/usr/bin/env python3
"""Randomised parity run on the GPU box: random sizes, contents, thresholds, options and batch sizes, product vs oracle.
Usage: tests/fuzz_parity.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from cudacam_amd import api, synth
api.preload_hip_runtime()
from oracle import oracle as O

O.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bad = 0
t0 = time.time()
for i in range(cases):
    w = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), rng.integers(240, 260), rng.integers(490, 510), rng.integers(600, 2200), rng.integers(2040, 8185)]))
    h = int(rng.choice([rng.integers(1, 12), rng.integers(12, 120), rng.integers(120, 400)]))
    kind = rng.choice(["noise", "natural", "flat", "steps", "sparse", "saturated"])
    seed = int(rng.integers(1, 1 << 30))
    if kind == "noise": img = synth.noise(w, h, seed)
    elif kind == "natural": img = synth.natural(w, h, seed)
    elif kind == "flat": img = synth.flat(w, h, int(rng.integers(0, 256)))
    elif kind == "steps": img = synth.steps(w, h, int(rng.integers(1, 256)), str(rng.choice(["vertical", "horizontal", "diagonal"])))
    elif kind == "sparse":
        img = np.zeros((h, w), np.uint8); k = max(1, w * h // 50)
        img[rng.integers(0, h, k), rng.integers(0, w, k)] = rng.integers(1, 256, k)
    else:
        img = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    mode = "R" if rng.random() < 0.75 else "O"
    low, high = sorted(int(x) for x in rng.integers(0, 256 if mode == "R" else 600, 2))
    nb = int(rng.integers(1, 4))
    frames = np.stack([np.roll(img, 3 * k, axis=1) for k in range(nb)])
    opts = {}
    if mode == "R":
        opts["sat"] = int(rng.integers(0, 2)); opts["split"] = int(rng.choice([2, 2, 2, 1, 0])); opts["chunk"] = int(rng.choice([0, 2, 8, 9, 24, 50, 122, 400]))   # split: 2 k_front8, 1 k_blur + k_nms, 0 k_front
        # round 3: the forms of k_front8 -- half-strip form and dense path: automatic, never, always
        opts["half"] = int(rng.choice([-1, -1, 0, 1, 1])); opts["dense"] = int(rng.choice([-1, -1, 0, 1]))
        # round 4: k_front_mx (blur and Sobel as i8 MFMAs) forced on in a third of the cases (it only takes one-channel k_front8 runs)
        opts["mx"] = int(rng.choice([0, 0, 1]))
        want = np.stack([O.canny_r(f, low, high, saturate=bool(opts["sat"])) for f in frames])
    else:
        opts["l2"] = int(rng.integers(0, 2)); opts["split"] = int(rng.choice([2, 2, 0])); opts["chunk"] = int(rng.choice([0, 2, 8, 9, 24, 50, 122, 400]))   # split: 2 k_front8o, 0 k_front_o
        want = np.stack([O.canny_o(f, low, high, l2gradient=bool(opts["l2"])) for f in frames])
    ch = 1
    if mode == "O" and rng.random() < 0.3:   # cv::Canny on 3-channel input: per pixel the channel with the largest magnitude
        ch = 3
        frames = np.stack([np.stack([np.roll(f, 5 * c, axis=0) ^ np.uint8(29 * c) for c in range(3)], axis=-1) for f in frames])
        opts["pc"] = 0
        want = np.stack([O.canny_o(f, low, high, l2gradient=bool(opts["l2"])) for f in frames])
    if mode == "R" and rng.random() < 0.3:   # interleaved 3-channel input: grey conversion (fused or fallback) or per-channel maps
        ch = 3
        frames = np.stack([np.stack([np.roll(f, 5 * c, axis=0) ^ np.uint8(17 * c) for c in range(3)], axis=-1) for f in frames])
        opts["pc"] = int(rng.integers(0, 2))
        if opts["pc"]:
            want = np.stack([O.canny_r(np.ascontiguousarray(f[:, :, c]), low, high, saturate=bool(opts["sat"])) for f in frames for c in range(3)])
        else:
            want = np.stack([O.canny_r(f, low, high, saturate=bool(opts["sat"])) for f in frames])
    with api.Context(w, h, ch, nb, api.MODE_R if mode == "R" else api.MODE_O, front_split=opts["split"]) as ctx:   # (split 1 / 0 in mode R: the test library)
        if ch == 3 and opts["pc"]:
            ctx.set_option(api.OPT_PER_CHANNEL, 1)
        ctx.set_thresholds(low, high)
        if mode == "R":
            ctx.set_option(api.OPT_NMS_SATURATE, opts["sat"]); ctx.set_option(api.OPT_FRONT_SPLIT, opts["split"]); ctx.set_tuning(opts["chunk"], 0)
            ctx.set_option(api.OPT_FRONT_HALF, opts["half"]); ctx.set_option(api.OPT_FRONT_DENSE, opts["dense"]); ctx.set_option(api.OPT_FRONT_MX, opts["mx"])
        else:
            ctx.set_option(api.OPT_L2_GRADIENT, opts["l2"]); ctx.set_option(api.OPT_FRONT_SPLIT, opts["split"]); ctx.set_tuning(opts["chunk"], 0)
        taps = mode == "R" and rng.random() < 0.3
        if taps:
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        n_in = ctx.upload(frames)
        ctx.run(api.CannyStage.HYSTER, n_in)
        got = ctx.download(len(want))
        if taps:   # the fast path's own blur and bit planes, stage by stage
            tb, tt = ctx.debug_tap(api.TAP_BLUR, len(want)), ctx.debug_tap(api.TAP_THRESH, len(want))
            planes = [np.ascontiguousarray(f[:, :, c]) for f in frames for c in range(3)] if ch == 3 and opts["pc"] else [O.gray_bgr(f) for f in frames] if ch == 3 else list(frames)
            for k, pl in enumerate(planes):
                st = O.canny_r(pl, low, high, stages=True, saturate=bool(opts["sat"]))
                if not (np.array_equal(tb[k], st["blur"]) and np.array_equal(tt[k], st["thresh"])):
                    bad += 1
                    print(f"MISMATCH (fast-path taps, map {k}) case {i}: {w}x{h} {kind} seed {seed} thr {low}/{high} nb {nb} ch {ch} {opts}", flush=True)
                    break
    if ch == 1 and rng.random() < 0.35:
        # the same case once more through the pipelined device path: three runs in a row (the frames rolled
        # differently each time) into two output buffers used in turn; the last two maps are checked
        import torch
        pitch = (w + 3) // 4 * 4
        def dev(fr):
            buf = np.zeros((nb, h, pitch), np.uint8); buf[:, :, :w] = fr
            return torch.from_numpy(buf).cuda()
        seq = [np.stack([np.roll(f, 7 * (r + 1), axis=0) for f in frames]) for r in range(3)]
        d_in = [dev(fr) for fr in seq]
        d_o = [torch.full((nb, h, pitch), 3, dtype=torch.uint8, device="cuda") for _ in range(2)]
        with api.Context(w, h, 1, nb, api.MODE_R if mode == "R" else api.MODE_O, front_split=opts["split"]) as ctx:
            ctx.set_thresholds(low, high)
            if mode == "R":
                ctx.set_option(api.OPT_NMS_SATURATE, opts["sat"]); ctx.set_option(api.OPT_FRONT_SPLIT, opts["split"]); ctx.set_tuning(opts["chunk"], 0)
                ctx.set_option(api.OPT_FRONT_HALF, opts["half"]); ctx.set_option(api.OPT_FRONT_DENSE, opts["dense"]); ctx.set_option(api.OPT_FRONT_MX, opts["mx"])
            else:
                ctx.set_option(api.OPT_L2_GRADIENT, opts["l2"])
            ctx.set_option(api.OPT_PIPELINE, 1)
            for r in range(3):
                ctx.run_device(d_in[r].data_ptr(), pitch, pitch * h, d_o[r % 2].data_ptr(), pitch, pitch * h, nb)
            ctx.sync()
        for r in (1, 2):
            g = d_o[r % 2].cpu().numpy()[:, :, :w]
            wnt = np.stack([O.canny_r(f, low, high, saturate=bool(opts["sat"])) if mode == "R" else O.canny_o(f, low, high, l2gradient=bool(opts["l2"])) for f in seq[r]])
            if not np.array_equal(g, wnt):
                bad += 1
                print(f"MISMATCH (pipelined run {r}) case {i}: {w}x{h} {kind} seed {seed} mode {mode} thr {low}/{high} nb {nb} {opts}", flush=True)
                break
    if not np.array_equal(got, want):
        bad += 1
        d = np.argwhere(got != want)
        print(f"MISMATCH case {i}: {w}x{h} {kind} seed {seed} mode {mode} thr {low}/{high} nb {nb} {opts}: {len(d)} px, first {d[0].tolist()}", flush=True)
print(f"fuzz: {cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
