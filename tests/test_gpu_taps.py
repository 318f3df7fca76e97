"""The FAST path's own intermediates against the oracle, bit for bit (run on the MI355X box, -m gpu).

test_gpu_parity.py::test_all_stages_mono checks the plain per-stage kernels behind `finalStage` < HYSTER;
those are not what a HYSTER run executes.  Here the front kernels themselves (k_blur + k_nms, the fused kernel,
k_front_o) are read back through hc_debug_tap: the blur they computed and the STRONG / CANDIDATE bit planes they
hand to the hysteresis, compared with oracle.gaussian / oracle.threshold (reference gaussianFilter5x5 and
doubleThreshold outputs, src/cvp/cannyEdgeD.cu:72-118, 273-293) -- so a wrong blur byte or a wrong candidate bit
that no strong pixel reaches cannot hide behind the final edge map."""
import numpy as np
import pytest

from cudacam_amd import api, synth

from test_gpu_parity import _diff, _images, _nms_queue_patterns

pytestmark = pytest.mark.gpu

FRONT_FORMS = [("front8", 2), ("split", 1), ("fused4", 0)]


def _tap_images():
    yield from _images()
    for name, img in _nms_queue_patterns():
        yield "nmsq_" + name, img
    yield "natural_1000x260", synth.natural(1000, 260, 21)
    yield "noise_2100x70", synth.noise(2100, 70, 22)          # more than one hysteresis panel wide


def _want(oracle, img, low, high, saturate=False):
    st = oracle.canny_r(img, low, high, stages=True, saturate=saturate)
    return st["blur"], st["thresh"], st["edges"]


@pytest.mark.parametrize("form,split", FRONT_FORMS, ids=[f for f, _ in FRONT_FORMS])
@pytest.mark.parametrize("name,img", list(_tap_images()), ids=[n for n, _ in _tap_images()])
def test_front_taps_mono(oracle, name, img, form, split):
    h, w = img.shape
    blur, thr, edges = _want(oracle, img, 10, 40)
    with api.Context(w, h, 1, 1, front_split=split) as ctx:
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)[0]
        _diff(ctx.debug_tap(api.TAP_BLUR)[0], blur, f"{name} {form}: blur of the front kernels")
        _diff(ctx.debug_tap(api.TAP_THRESH)[0], thr, f"{name} {form}: bit planes of the front kernels")
        _diff(got, edges, f"{name} {form}: edges")


@pytest.mark.parametrize("form,split", FRONT_FORMS, ids=[f for f, _ in FRONT_FORMS])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_front_taps_thresholds_and_batch(oracle, form, split, pipeline):
    """Other thresholds, the saturating NMS variant, a batch, pipelined mode (provisional map on)."""
    frames = np.stack([synth.natural(520, 300, 31), synth.noise(520, 300, 32), synth.steps(520, 300, 250, "diagonal")])
    for low, high, sat in ((10, 40, 0), (60, 200, 0), (0, 255, 0), (25, 25, 1)):
        with api.Context(520, 300, 1, 3, front_split=split) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_NMS_SATURATE, sat)
            ctx.set_option(api.OPT_PIPELINE, pipeline)
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
            got = ctx.process(frames)
            tb, tt = ctx.debug_tap(api.TAP_BLUR, 3), ctx.debug_tap(api.TAP_THRESH, 3)
            for f in range(3):
                blur, thr, edges = _want(oracle, frames[f], low, high, bool(sat))
                tag = f"frame {f} {form} {low}/{high} sat={sat} pipeline={pipeline}"
                _diff(tb[f], blur, tag + ": blur")
                _diff(tt[f], thr, tag + ": bit planes")
                _diff(got[f], edges, tag + ": edges")


@pytest.mark.parametrize("form,split", FRONT_FORMS, ids=[f for f, _ in FRONT_FORMS])
def test_front_taps_bgr_and_per_channel(oracle, form, split):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (130, 501, 3), dtype=np.uint8)
    img[40:90, 100:300] = (200, 30, 90)
    mono = oracle.gray_bgr(img)
    blur, thr, edges = _want(oracle, mono, 10, 40)
    with api.Context(501, 130, 3, 1, front_split=split) as ctx:  # BGR -> grey fused into the front kernel's load
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)[0]
        _diff(ctx.debug_tap(api.TAP_BLUR)[0], blur, f"bgr {form}: blur")
        _diff(ctx.debug_tap(api.TAP_THRESH)[0], thr, f"bgr {form}: bit planes")
        _diff(got, edges, f"bgr {form}: edges")
    with api.Context(501, 130, 3, 1, front_split=split) as ctx:  # one map per channel: output frame 3 f + ch
        ctx.set_option(api.OPT_PER_CHANNEL, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        n = ctx.upload(img)
        ctx.run(api.CannyStage.HYSTER, n)
        got = ctx.download(3)
        tb, tt = ctx.debug_tap(api.TAP_BLUR, 3), ctx.debug_tap(api.TAP_THRESH, 3)
        for ch in range(3):
            blur, thr, edges = _want(oracle, np.ascontiguousarray(img[:, :, ch]), 10, 40)
            _diff(tb[ch], blur, f"channel {ch} {form}: blur")
            _diff(tt[ch], thr, f"channel {ch} {form}: bit planes")
            _diff(got[ch], edges, f"channel {ch} {form}: edges")


@pytest.mark.parametrize("form", ["front8o", "front_o"])
@pytest.mark.parametrize("l2", [0, 1])
@pytest.mark.parametrize("name,img", list(_images()), ids=[n for n, _ in _images()])
def test_front_taps_mode_o(oracle, name, img, l2, form):
    """Mode O: the bit planes the front kernel hands to the hysteresis equal cv::Canny's map before its flood (the CPU
    restatement's), for the 8-px k_front8o (default) and the 4-px k_front_o."""
    h, w = img.shape
    low, high = (50, 150) if not l2 else (40, 120)
    edges, pre = oracle.canny_o_stages(img, low, high, bool(l2))
    with api.Context(w, h, 1, 1, mode=api.MODE_O) as ctx:
        ctx.set_thresholds(low, high)
        ctx.set_option(api.OPT_L2_GRADIENT, l2)
        ctx.set_option(api.OPT_FRONT_SPLIT, 2 if form == "front8o" else 0)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)[0]
        assert ctx.last_run_info()[2] == (3 if form == "front8o" else -1)
        _diff(ctx.debug_tap(api.TAP_THRESH)[0], pre, f"{name} mode O l2={l2}: bit planes of {form}")
        _diff(got, edges, f"{name} mode O l2={l2}: edges")
        with pytest.raises(api.HipCannyError):
            ctx.debug_tap(api.TAP_BLUR)


def test_tap_needs_option():
    with api.Context(64, 48, 1, 1) as ctx:
        ctx.process(synth.noise(64, 48, 1))
        with pytest.raises(api.HipCannyError):
            ctx.debug_tap(api.TAP_THRESH)


def _bgr_images():
    rng = np.random.default_rng(8)
    yield "noise_333x222", rng.integers(0, 256, (222, 333, 3), dtype=np.uint8)
    yield "natural3_500x260", np.stack([synth.natural(500, 260, 70 + c) for c in range(3)], axis=-1)
    a = np.stack([synth.natural(249, 130, 90)] * 3, axis=-1).copy()      # equal channels: every pixel is a tie -> channel 0
    yield "ties_249x130", a
    b = np.zeros((64, 96, 3), np.uint8)
    b[:, 40:, 0] = 200; b[20:, :, 1] = 120; b[:, :, 2] = np.arange(96, dtype=np.uint8)[None, :] * 2   # a different winner per region
    yield "regions_96x64", b
    yield "tiny_5x3", rng.integers(0, 256, (3, 5, 3), dtype=np.uint8)


@pytest.mark.parametrize("l2", [0, 1])
@pytest.mark.parametrize("name,img", list(_bgr_images()), ids=[n for n, _ in _bgr_images()])
def test_mode_o_three_channel(oracle, name, img, l2):
    """cv::Canny on 3-channel input: per pixel the derivatives of the channel with the largest magnitude, first on ties
    (oracle: orc_canny_o channels = 3).  Checked before the flood (bit planes) and after it."""
    h, w = img.shape[:2]
    low, high = (50, 150) if not l2 else (40, 120)
    edges, pre = oracle.canny_o_stages(img, low, high, bool(l2))
    for pipeline in (0, 1):
        with api.Context(w, h, 3, 2, mode=api.MODE_O) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_L2_GRADIENT, l2)
            ctx.set_option(api.OPT_PIPELINE, pipeline)
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
            got = ctx.process(np.stack([img, img[::-1].copy()]))
            _diff(ctx.debug_tap(api.TAP_THRESH, 2)[0], pre, f"{name} mode O 3-channel l2={l2}: bit planes of k_front_o")
            _diff(got[0], edges, f"{name} mode O 3-channel l2={l2} pipeline={pipeline}: edges")
            _diff(got[1], oracle.canny_o(img[::-1].copy(), low, high, bool(l2)), f"{name} flipped")


def test_mode_o_three_channel_device_buffers(oracle):
    """Tight 3-channel device rows that do not hold whole 12-byte groups are staged through the internal buffer."""
    import torch
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (77, 101, 3), dtype=np.uint8)   # pitch 303: not a multiple of 4
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.zeros((77, 101), dtype=torch.uint8, device="cuda")
    with api.Context(101, 77, 3, 1, mode=api.MODE_O) as ctx:
        torch.cuda.synchronize()
        ctx.run_device(d_in.data_ptr(), 303, 303 * 77, d_out.data_ptr(), 101, 101 * 77, 1)
        ctx.sync()
    _diff(d_out.cpu().numpy(), oracle.canny_o(img, 50, 150), "mode O 3-channel, unaligned device buffers")


@pytest.mark.parametrize("w,h,nb,form", [(640, 480, 400, 4), (640, 480, 3, 4), (640, 480, 1, 2), (200, 120, 4200, 4), (744, 60, 2300, 2), (745, 60, 2300, 2), (496, 60, 3400, 2), (1000, 60, 2300, 4), (1280, 90, 900, 2)])
def test_default_form_by_width(oracle, w, h, nb, form):
    """The library picks k_front8's form by how many waves a run needs: the half-strip form (two 240-column half-waves per
    wave, units paired across strips and frames: form 4) whenever that is fewer than with 496-column strips (form 2) --
    640 columns: 1.5 waves per frame instead of 2 -- and never one of the round-1 4-px kernels (forms 1 / 0: round 2 sent
    narrow big batches there).  The same blur, bit planes and edges either way."""
    uniq = np.stack([synth.natural(w, h, 5 + w + k) for k in range(4)])
    frames = np.tile(uniq, ((nb + 3) // 4, 1, 1))[:nb]
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(frames)
        assert ctx.last_run_info()[2] == form
        tb, tt = ctx.debug_tap(api.TAP_BLUR, nb), ctx.debug_tap(api.TAP_THRESH, nb)
        for k in sorted({0, 1, 2, 3, nb - 1} & set(range(nb))):
            blur, thr, edges = _want(oracle, frames[k], 10, 40)
            _diff(tb[k], blur, f"{w}x{h} frame {k}: blur")
            _diff(tt[k], thr, f"{w}x{h} frame {k}: bit planes")
            _diff(got[k], edges, f"{w}x{h} frame {k}: edges")
