"""BASELINE.json's full-size configurations on the MI355X (-m gpu), checked against the oracle bit for bit:
configs[4] 7680x4320 three-channel per-channel Canny, and an 8K mono frame; plus the fast path's own blur and bit
planes at that size (hc_debug_tap).  The oracle takes about a second per 8K plane on one host thread."""
import numpy as np
import pytest

from cudacam_amd import api, synth

from test_gpu_parity import _diff

pytestmark = pytest.mark.gpu
W8K, H8K = 7680, 4320


def _plane(seed, kind):
    if kind == "noise":
        return synth.noise(W8K, H8K, seed)
    # natural() draws its shapes in Python: build the 8K plane from 1080p tiles of different seeds (4 x 4)
    tiles = [[synth.natural(1920, 1080, seed * 16 + 4 * r + c) for c in range(4)] for r in range(4)]
    return np.block(tiles)


def test_8k_mono_matches_oracle(oracle):
    img = _plane(3, "natural")
    assert img.shape == (H8K, W8K)
    st = oracle.canny_r(img, 10, 40, stages=True)
    with api.Context(W8K, H8K, 1, 1) as ctx:
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)[0]
        _diff(ctx.debug_tap(api.TAP_BLUR)[0], st["blur"], "8K mono: blur of the front kernels")
        _diff(ctx.debug_tap(api.TAP_THRESH)[0], st["thresh"], "8K mono: bit planes of the front kernels")
        _diff(got, st["edges"], "8K mono: edges")
        work, _ = ctx.hysteresis_info()
        assert work >= 1


@pytest.mark.parametrize("pipeline", [0, 1])
def test_8k_three_channel_per_channel_matches_oracle(oracle, pipeline):
    """BASELINE configs[4]: 7680x4320 interleaved 3-channel input, one Canny map per channel, in one run."""
    planes = [_plane(5, "natural"), _plane(6, "noise"), np.ascontiguousarray(_plane(7, "natural")[::-1, ::-1])]
    img = np.ascontiguousarray(np.stack(planes, axis=-1))
    assert img.shape == (H8K, W8K, 3)
    want = oracle.canny_r_batch(np.stack(planes), 10, 40, threads=3)
    with api.Context(W8K, H8K, 3, 1) as ctx:
        ctx.set_option(api.OPT_PER_CHANNEL, 1)
        ctx.set_option(api.OPT_PIPELINE, pipeline)
        n = ctx.upload(img)
        ctx.run(api.CannyStage.HYSTER, n)
        got = ctx.download(3)
    for ch in range(3):
        _diff(got[ch], want[ch], f"8K per-channel, channel {ch}, pipeline={pipeline}")


def test_4k_batch_matches_oracle(oracle):
    """BASELINE configs[2]: 3840x2160 grey, a small batch."""
    frames = np.stack([np.block([[synth.natural(1920, 1080, 40 + 4 * f + 2 * r + c) for c in range(2)] for r in range(2)]) for f in range(3)])
    want = oracle.canny_r_batch(frames, 10, 40, threads=3)
    with api.Context(3840, 2160, 1, 3) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        got = ctx.process(frames)
    for f in range(3):
        _diff(got[f], want[f], f"4K frame {f}")


@pytest.mark.parametrize("l2", [0, 1])
def test_8k_mode_o_matches_restatement(oracle, l2):
    """cv::Canny semantics at full size: k_front8o (8 px per lane, id-only queue, replicate border in the batch) on an
    8K frame and on BASELINE configs[1]'s 1920x1080, bit planes before the flood and final map, both gradient norms."""
    for img, tag in ((_plane(9, "natural"), "8K"), (synth.natural(1920, 1080, 17), "1080p")):
        low, high = (50, 150) if not l2 else (40, 120)
        edges, pre = oracle.canny_o_stages(img, low, high, bool(l2))
        h, w = img.shape
        with api.Context(w, h, 1, 1, mode=api.MODE_O) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_L2_GRADIENT, l2)
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
            got = ctx.process(img)[0]
            assert ctx.last_run_info()[2] == 3
            _diff(ctx.debug_tap(api.TAP_THRESH)[0], pre, f"{tag} mode O l2={l2}: bit planes of k_front8o")
            _diff(got, edges, f"{tag} mode O l2={l2}: edges")
