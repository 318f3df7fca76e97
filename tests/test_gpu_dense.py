"""k_front8's dense path (round 3): windows in which many half-lanes pass the low threshold are processed by wave-wide
non-maximum suppression in registers instead of the id queue and its batches (front8.hip, dense_window).  Both paths
must produce the same bit planes.  Forced on for every window (HC_OPT_FRONT_DENSE = 1) over the whole tap-test image
set -- borders, flat frames, the u8 wrap bands of strong steps, noise, one-pixel frames -- and in the automatic mode on
frames whose windows alternate between the two paths; the fast path's own bit planes (hc_debug_tap) and the final maps
are compared with the oracle, bit for bit."""
import numpy as np
import pytest

from cudacam_amd import api, synth

from test_gpu_parity import _diff
from test_gpu_taps import _tap_images, _want

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,img", list(_tap_images()), ids=[n for n, _ in _tap_images()])
def test_dense_every_window_mono(oracle, name, img):
    h, w = img.shape
    blur, thr, edges = _want(oracle, img, 10, 40)
    with api.Context(w, h, 1, 1) as ctx:
        ctx.set_option(api.OPT_FRONT_DENSE, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)[0]
        _diff(ctx.debug_tap(api.TAP_BLUR)[0], blur, f"{name} dense: blur")
        _diff(ctx.debug_tap(api.TAP_THRESH)[0], thr, f"{name} dense: bit planes")
        _diff(got, edges, f"{name} dense: edges")


def _mixed(w, h, seed):
    """Bands of noise between natural content: the windows of a run switch between the paths (and back)."""
    img = synth.natural(w, h, seed)
    nz = synth.noise(w, h, seed + 1)
    for y0 in range(20, h - 30, 97):
        img[y0:y0 + 31] = nz[y0:y0 + 31]
    img[:, w // 3: w // 3 + 40] = nz[:, w // 3: w // 3 + 40]
    return img


@pytest.mark.parametrize("mode", [-1, 0, 1])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_dense_modes_thresholds_and_batch(oracle, mode, pipeline):
    frames = np.stack([_mixed(1000, 300, 31), synth.noise(1000, 300, 32), synth.steps(1000, 300, 250, "diagonal"), synth.natural(1000, 300, 33)])
    for low, high, sat in ((10, 40, 0), (60, 200, 0), (0, 255, 0), (25, 25, 1)):
        with api.Context(1000, 300, 1, 4) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_NMS_SATURATE, sat)
            ctx.set_option(api.OPT_FRONT_DENSE, mode)
            ctx.set_option(api.OPT_PIPELINE, pipeline)
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
            got = ctx.process(frames)
            thr = ctx.debug_tap(api.TAP_THRESH, 4)
            for f in range(4):
                st = oracle.canny_r(frames[f], low, high, stages=True, saturate=bool(sat))
                _diff(thr[f], st["thresh"], f"dense mode {mode}, pipeline {pipeline}, {low}/{high}/{sat}, frame {f}: bit planes")
                _diff(got[f], st["edges"], f"dense mode {mode}, pipeline {pipeline}, {low}/{high}/{sat}, frame {f}: edges")


@pytest.mark.parametrize("chunk", [8, 26, 120])
@pytest.mark.parametrize("w", [96, 640, 1000, 2100])
def test_dense_run_lengths_widths_and_half_strips(oracle, w, chunk):
    """Run lengths (the first / last windows of a run, rows that belong to the neighbouring runs), several strips, the
    half-strip form, frames wider than one hysteresis panel."""
    h, n = 200, 3
    frames = np.stack([synth.noise(w, h, 70 + w), _mixed(w, h, 71 + w) if w >= 200 else synth.natural(w, h, 71 + w), synth.steps(w, h, 255, "vertical")])
    want = [oracle.canny_r(f, 10, 40, stages=True) for f in frames]
    with api.Context(w, h, 1, n) as ctx:
        ctx.set_option(api.OPT_FRONT_DENSE, 1)
        ctx.set_option(api.OPT_FRONT_HALF, 1 if w <= 1000 else -1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        ctx.set_tuning(chunk, 0)
        got = ctx.process(frames)
        thr = ctx.debug_tap(api.TAP_THRESH, n)
        for f in range(n):
            _diff(thr[f], want[f]["thresh"], f"{w} wide, runs of {chunk}, frame {f}: bit planes")
            _diff(got[f], want[f]["edges"], f"{w} wide, runs of {chunk}, frame {f}: edges")


@pytest.mark.parametrize("per_channel", [0, 1])
def test_dense_three_channel(oracle, per_channel):
    w, h, n = 520, 150, 2
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    img[1, :, : w // 2] = np.stack([synth.natural(w // 2, h, 5 + c) for c in range(3)], axis=-1)
    with api.Context(w, h, 3, n) as ctx:
        ctx.set_option(api.OPT_PER_CHANNEL, per_channel)
        ctx.set_option(api.OPT_FRONT_DENSE, 1)
        got = ctx.process(img)
        for f in range(n):
            if per_channel:
                for c in range(3):
                    _diff(got[3 * f + c], oracle.canny_r(np.ascontiguousarray(img[f, :, :, c]), 10, 40), f"dense per-channel frame {f} channel {c}")
            else:
                _diff(got[f], oracle.canny_r(img[f], 10, 40), f"dense BGR frame {f}")


def test_dense_full_size_noise_batch(oracle):
    """1080p noise, automatic mode, device-resident and pipelined: the content the dense path exists for."""
    import torch
    w, h, nb = 1920, 1080, 6
    frames = np.stack([synth.noise(w, h, 500 + f) if f % 3 else _mixed(w, h, 500 + f) for f in range(nb)])
    want = oracle.canny_r_batch(frames, 10, 40, threads=8)
    d_in = torch.from_numpy(frames).cuda()
    d_out = [torch.zeros_like(d_in) for _ in range(2)]
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        for r in range(3):
            ctx.run_device(d_in.data_ptr(), w, w * h, d_out[r % 2].data_ptr(), w, w * h, nb)
        ctx.sync()
        for o in d_out:
            got = o.cpu().numpy()
            for f in range(nb):
                _diff(got[f], want[f], f"1080p dense batch, frame {f}")
