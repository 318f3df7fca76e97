"""k_front_mx (round 4): the front path of Mode R with the 5x5 Gaussian sum and the 3x3 Sobel sums computed by
v_mfma_i32_32x32x32_i8 (strips of 216 columns, blocks of 16 rows, LDS rings of 32-byte tile segments).  Forced on
(HC_OPT_FRONT_MX = 1) and checked like the other forms: the fast path's own blur (hc_debug_tap: every byte the matrix
pipe and the fix-up produced) and bit planes, and the final maps, against the oracle, bit for bit -- every tap-test
image (borders, flat frames, the wrap bands, 1 x 1), widths around the strip and tile boundaries, run lengths,
thresholds, the saturating variant, batches, pipelined mode with the provisional map, big batches, one-wave workgroups."""
import numpy as np
import pytest

from cudacam_amd import api, synth

from test_gpu_parity import _diff
from test_gpu_taps import _tap_images

pytestmark = pytest.mark.gpu

# strips of 216 columns, tiles of 28: widths at and around their boundaries
WIDTHS = [4, 27, 28, 29, 56, 200, 215, 216, 217, 220, 224, 244, 432, 433, 640, 1000, 1296, 1920]


def _frames(w, h, n, seed):
    out = []
    for f in range(n):
        k = (seed + f) % 4
        out.append(synth.natural(w, h, seed + f) if k == 0 else synth.noise(w, h, seed + f) if k == 1
                   else synth.steps(w, h, 250 - 3 * f, ("vertical", "horizontal", "diagonal")[f % 3]) if k == 2 else synth.serpentine(w, h) if w >= 200 and h >= 80 else synth.flat(w, h, 90 + f))
    return np.stack(out)


def _check(oracle, ctx, frames, got, low=10, high=40, sat=False, tag=""):
    n = len(frames)
    assert ctx.last_run_info()[2] == 5, "k_front_mx did not run"
    blur, thr = ctx.debug_tap(api.TAP_BLUR, n), ctx.debug_tap(api.TAP_THRESH, n)
    for f in range(n):
        st = oracle.canny_r(frames[f], low, high, stages=True, saturate=sat)
        _diff(blur[f], st["blur"], f"{tag} frame {f} of {n}: blur")
        _diff(thr[f], st["thresh"], f"{tag} frame {f} of {n}: bit planes")
        _diff(got[f], st["edges"], f"{tag} frame {f} of {n}: edges")


@pytest.mark.parametrize("name,img", list(_tap_images()), ids=[n for n, _ in _tap_images()])
def test_mx_taps_mono(oracle, name, img):
    h, w = img.shape
    with api.Context(w, h, 1, 1) as ctx:
        ctx.set_option(api.OPT_FRONT_MX, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)
        _check(oracle, ctx, img[None], got, tag=name)


@pytest.mark.parametrize("n", [1, 3])
@pytest.mark.parametrize("w", WIDTHS)
def test_mx_widths(oracle, w, n):
    h = 83
    frames = _frames(w, h, n, 40 + w)
    with api.Context(w, h, 1, n) as ctx:
        ctx.set_option(api.OPT_FRONT_MX, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(frames)
        _check(oracle, ctx, frames, got, tag=f"{w}x{h}")


@pytest.mark.parametrize("h", [1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 130])
def test_mx_heights(oracle, h):
    w, n = 300, 2
    frames = _frames(w, h, n, 11 + h)
    with api.Context(w, h, 1, n) as ctx:
        ctx.set_option(api.OPT_FRONT_MX, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(frames)
        _check(oracle, ctx, frames, got, tag=f"{w}x{h}")


@pytest.mark.parametrize("chunk", [8, 16, 20, 50, 300])
def test_mx_run_lengths_and_thresholds(oracle, chunk):
    """Rows per work item, other thresholds and the saturating NMS variant never change the result."""
    w, h, n = 640, 230, 3
    frames = _frames(w, h, n, 7)
    for low, high, sat in ((10, 40, 0), (60, 200, 0), (0, 255, 0), (25, 25, 1)):
        with api.Context(w, h, 1, n) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_NMS_SATURATE, sat)
            ctx.set_option(api.OPT_FRONT_MX, 1)
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
            ctx.set_tuning(chunk, 0)
            got = ctx.process(frames)
            _check(oracle, ctx, frames, got, low, high, bool(sat), tag=f"chunk {chunk}, thresholds {low}/{high}, saturate {sat}")


def test_mx_pipelined_device_buffers(oracle):
    """Device-resident batches, pipelined (the front kernel writes the provisional map; the hysteresis patches it), two
    output buffers in turn; k_front_mx, then k_front8 in the same context: identical maps."""
    import torch
    w, h, nb = 640, 480, 5
    runs = [_frames(w, h, nb, 100 + 11 * r) for r in range(4)]
    want = [oracle.canny_r_batch(b, 10, 40, threads=4) for b in runs]
    d_in = [torch.from_numpy(b).cuda() for b in runs]
    d_out = [torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda") for _ in range(4)]
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        for mode, form in ((1, 5), (0, 2), (1, 5)):
            ctx.set_option(api.OPT_FRONT_MX, mode)
            ctx.set_option(api.OPT_FRONT_HALF, 0)
            for r in range(4):
                ctx.run_device(d_in[r].data_ptr(), w, w * h, d_out[r].data_ptr(), w, w * h, nb)
            ctx.sync()
            assert ctx.last_run_info() == (False, False, form)
            for r in range(4):
                got = d_out[r].cpu().numpy()
                for f in range(nb):
                    _diff(got[f], want[r][f], f"HC_OPT_FRONT_MX {mode}, run {r}, frame {f}")
                d_out[r].zero_()


@pytest.mark.parametrize("w", [640, 1920])
def test_mx_big_batches_and_default(oracle, w):
    """Runs of 0.13 G pixels (pipelined, three output buffers in turn, long runs of rows): k_front_mx on request, k_front8
    by default; the maps are the same."""
    import torch
    h = 120
    nb = 130_000_000 // (w * h) + 1
    uniq = _frames(w, h, 6, 300 + w)
    want = np.stack([oracle.canny_r(f, 10, 40) for f in uniq])
    reps = (nb + 5) // 6
    d_in = torch.from_numpy(np.tile(uniq, (reps, 1, 1))[:nb].copy()).cuda()
    d_out = [torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda") for _ in range(3)]
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        ctx.set_option(api.OPT_FRONT_HALF, 0)   # (640 columns: the half-strip form of k_front8 would go first)
        for mode, form in ((None, 2), (1, 5), (0, 2)):
            if mode is not None:
                ctx.set_option(api.OPT_FRONT_MX, mode)
            for r in range(5):
                ctx.run_device(d_in.data_ptr(), w, w * h, d_out[r % 3].data_ptr(), w, w * h, nb)
            ctx.sync()
            assert ctx.last_run_info()[2] == form
            for r in range(3):
                got = d_out[r].cpu().numpy()
                for f in range(nb):
                    if not np.array_equal(got[f], want[f % 6]):
                        _diff(got[f], want[f % 6], f"{w}x{h}, HC_OPT_FRONT_MX {mode}, buffer {r}, frame {f}")
                d_out[r].zero_()


@pytest.mark.parametrize("wpb", [1, 4])
def test_mx_waves_per_workgroup(oracle, wpb):
    """One-wave and four-wave workgroups (HC_OPT_FRONT_WPB) compute the same maps."""
    w, h, n = 1296, 300, 4
    frames = _frames(w, h, n, 77)
    with api.Context(w, h, 1, n) as ctx:
        ctx.set_option(api.OPT_FRONT_MX, 1)
        ctx.set_option(api.OPT_FRONT_WPB, wpb)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(frames)
        _check(oracle, ctx, frames, got, tag=f"{wpb} waves per workgroup")
        assert ctx.front_waves_per_workgroup() == wpb
