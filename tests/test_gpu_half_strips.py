"""k_front8's half-strip form (narrow frames; round 3): a wave is two independent half-waves of 240 columns, and the
(frame, half-strip) units of a run of rows are dealt to them in pairs -- so the two halves of a wave may work on
different strips of one frame or on different frames.  Forced on (HC_OPT_FRONT_HALF = 1) at widths where every pairing
occurs, and checked like the plain form: the fast path's own blur and bit planes (hc_debug_tap) and the final maps
against the oracle, bit for bit; mono, BGR -> grey, per-channel; batches whose unit count is odd (a half-wave without a
unit); pipelined mode with the provisional map; the automatic choice at 640 x 480 (the reference's webcam format,
src/io/webcam.cpp:39-40, BASELINE configs[0] size)."""
import numpy as np
import pytest

from cudacam_amd import api, synth

from test_gpu_parity import _diff

pytestmark = pytest.mark.gpu

# width -> half-strips: 96 (1), 240 (1, exact), 248 (2), 480 (2, exact), 500 (3), 640 (3), 720 (3, exact), 1000 (5), 1210 (6)
WIDTHS = [96, 240, 248, 480, 500, 640, 720, 1000, 1210]


def _frames(w, h, n, seed):
    out = []
    for f in range(n):
        k = (seed + f) % 4
        out.append(synth.natural(w, h, seed + f) if k == 0 else synth.noise(w, h, seed + f) if k == 1
                   else synth.steps(w, h, 250 - 3 * f, ("vertical", "horizontal", "diagonal")[f % 3]) if k == 2 else synth.serpentine(w, h) if w >= 200 and h >= 80 else synth.flat(w, h, 90 + f))
    return np.stack(out)


@pytest.mark.parametrize("n", [1, 2, 3, 5])
@pytest.mark.parametrize("w", WIDTHS)
def test_half_form_mono_taps_and_edges(oracle, w, n):
    h = 83
    frames = _frames(w, h, n, 40 + w)
    with api.Context(w, h, 1, n) as ctx:
        ctx.set_option(api.OPT_FRONT_HALF, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(frames)
        assert ctx.last_run_info()[2] == 4, "the half-strip form did not run"
        blur, thr = ctx.debug_tap(api.TAP_BLUR, n), ctx.debug_tap(api.TAP_THRESH, n)
        for f in range(n):
            st = oracle.canny_r(frames[f], 10, 40, stages=True)
            _diff(blur[f], st["blur"], f"{w}x{h} frame {f} of {n}: blur")
            _diff(thr[f], st["thresh"], f"{w}x{h} frame {f} of {n}: bit planes")
            _diff(got[f], st["edges"], f"{w}x{h} frame {f} of {n}: edges")


@pytest.mark.parametrize("chunk", [8, 20, 50, 300])
def test_half_form_run_lengths_and_thresholds(oracle, chunk):
    """Rows per work item, other thresholds and the saturating NMS variant never change the result."""
    w, h, n = 640, 230, 3
    frames = _frames(w, h, n, 7)
    for low, high, sat in ((10, 40, 0), (60, 200, 0), (0, 255, 0), (25, 25, 1)):
        want = [oracle.canny_r(f, low, high, saturate=bool(sat)) for f in frames]
        with api.Context(w, h, 1, n) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_NMS_SATURATE, sat)
            ctx.set_option(api.OPT_FRONT_HALF, 1)
            ctx.set_tuning(chunk, 0)
            got = ctx.process(frames)
            assert ctx.last_run_info()[2] == 4
            for f in range(n):
                _diff(got[f], want[f], f"chunk {chunk}, thresholds {low}/{high}, saturate {sat}, frame {f}")


@pytest.mark.parametrize("per_channel", [0, 1])
@pytest.mark.parametrize("w", [96, 500, 640, 1000])
def test_half_form_three_channel(oracle, w, per_channel):
    h, n = 70, 3
    rng = np.random.default_rng(5 + w)
    img = np.stack([np.stack([synth.natural(w, h, 60 + 3 * f + c) for c in range(3)], axis=2) for f in range(n)])
    img[1, 10:40, w // 5: w // 2, 2] = rng.integers(0, 256, (30, w // 2 - w // 5), dtype=np.uint8)
    with api.Context(w, h, 3, n) as ctx:
        ctx.set_option(api.OPT_PER_CHANNEL, per_channel)
        ctx.set_option(api.OPT_FRONT_HALF, 1)
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        got = ctx.process(img)
        assert ctx.last_run_info()[2] == 4
        nmaps = 3 * n if per_channel else n
        assert got.shape[0] == nmaps
        thr = ctx.debug_tap(api.TAP_THRESH, nmaps)
        for f in range(n):
            if per_channel:
                for c in range(3):
                    st = oracle.canny_r(np.ascontiguousarray(img[f, :, :, c]), 10, 40, stages=True)
                    _diff(thr[3 * f + c], st["thresh"], f"{w} per-channel frame {f} channel {c}: bit planes")
                    _diff(got[3 * f + c], st["edges"], f"{w} per-channel frame {f} channel {c}: edges")
            else:
                st = oracle.canny_r(img[f], 10, 40, stages=True)
                _diff(thr[f], st["thresh"], f"{w} BGR frame {f}: bit planes")
                _diff(got[f], st["edges"], f"{w} BGR frame {f}: edges")


def test_half_form_pipelined_device_buffers(oracle):
    """Device-resident batches, pipelined (the front kernel writes the provisional map; the hysteresis patches it), two
    output buffers in turn; then the same context with the plain form: identical maps."""
    import torch
    w, h, nb = 640, 480, 5
    runs = [_frames(w, h, nb, 100 + 11 * r) for r in range(4)]
    want = [oracle.canny_r_batch(b, 10, 40, threads=4) for b in runs]
    d_in = [torch.from_numpy(b).cuda() for b in runs]
    d_out = [torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda") for _ in range(4)]
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        for mode, form in ((-1, 4), (0, 2), (1, 4)):   # automatic: 5 frames x 3 half-strips = 8 waves instead of 10
            ctx.set_option(api.OPT_FRONT_HALF, mode)
            for r in range(4):
                ctx.run_device(d_in[r].data_ptr(), w, w * h, d_out[r].data_ptr(), w, w * h, nb)
            ctx.sync()
            assert ctx.last_run_info() == (False, False, form)
            for r in range(4):
                got = d_out[r].cpu().numpy()
                for f in range(nb):
                    _diff(got[f], want[r][f], f"HC_OPT_FRONT_HALF {mode}, run {r}, frame {f}")
                d_out[r].zero_()


@pytest.mark.parametrize("w,channels", [(640, 1), (1000, 1), (1920, 1), (1000, 3)])
def test_one_wave_workgroups_pipelined(oracle, w, channels):
    """HC_OPT_FRONT_WPB: k_front8 in one-wave workgroups (pipelined big batches, mono / BGR) gives the maps of the
    four-wave form -- plain and half-strip form, device-resident, three output buffers in turn; 0.5 G pixels per run make
    it a big batch (the small ones keep four waves)."""
    import torch
    h = 120
    nb = 500_000_000 // (w * h) + 1
    uniq = _frames(w, h, 6, 300 + w) if channels == 1 else np.stack([np.stack([synth.natural(w, h, 400 + 3 * f + c) for c in range(3)], axis=2) for f in range(6)])
    want = np.stack([oracle.canny_r(f, 10, 40) for f in uniq])
    reps = (nb + 5) // 6
    d_in = torch.from_numpy(np.tile(uniq, (reps,) + (1,) * (uniq.ndim - 1))[:nb].copy()).cuda()
    d_out = [torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda") for _ in range(3)]
    with api.Context(w, h, channels, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        for mode, waves in ((1, 1), (4, 4), (-1, None)):
            ctx.set_option(api.OPT_FRONT_WPB, mode)
            for r in range(7):
                ctx.run_device(d_in.data_ptr(), w * channels, w * channels * h, d_out[r % 3].data_ptr(), w, w * h, nb)
            ctx.sync()
            if waves is not None:
                assert ctx.front_waves_per_workgroup() == waves
            else:
                assert ctx.front_waves_per_workgroup() in (1, 4)
            for o in d_out:
                got = o[:: max(1, nb // 24)].cpu().numpy()
                for j in range(got.shape[0]):
                    _diff(got[j], want[(j * max(1, nb // 24)) % 6], f"HC_OPT_FRONT_WPB {mode}, {w} wide, {channels} channel(s), frame {j * max(1, nb // 24)}")
                o.zero_()
        with pytest.raises(api.HipCannyError):
            ctx.set_option(api.OPT_FRONT_WPB, 2)


def test_ragged_tight_rows_are_staged_onto_the_8px_kernel(oracle):
    """Tight rows whose width is not a multiple of 8 used to fall back to the 4-px kernels; they are staged through the
    internal pitched buffer instead (hc_last_run_info says so) and every frame stays on the one-kernel path."""
    import torch
    w, h, nb = 644, 60, 2
    frames = _frames(w, h, nb, 9)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    with api.Context(w, h, 1, nb) as ctx:
        ctx.run_device(d_in.data_ptr(), w, w * h, d_out.data_ptr(), w, w * h, nb)
        ctx.sync()
        staged_in, staged_out, form = ctx.last_run_info()
        assert staged_in and not staged_out and form in (2, 4)
        got = d_out.cpu().numpy()
        for f in range(nb):
            _diff(got[f], oracle.canny_r(frames[f], 10, 40), f"staged ragged rows, frame {f}")
