"""Property tests (hypothesis), SURVEY 4 item 4: size-independent laws of the pipeline.

CPU (not gpu): the oracle obeys them.  GPU (-m gpu): the HIP path obeys them on random shapes and at BASELINE's full
1080p size, where the oracle is only consulted on a sample."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from cudacam_amd import synth

SET = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def _img(seed, w, h, kind):
    return synth.noise(w, h, seed) if kind == 0 else synth.natural(w, h, seed, nshapes=6) if kind == 1 else synth.steps(w, h, 40 + seed % 216, ("vertical", "horizontal", "diagonal")[seed % 3])


shape = st.tuples(st.integers(0, 10_000), st.integers(1, 96), st.integers(1, 64), st.integers(0, 2))


# ---- the oracle ------------------------------------------------------------------------------------------------------
@settings(max_examples=40, **SET)
@given(shape, st.integers(0, 255), st.integers(0, 255))
def test_oracle_threshold_laws(oracle, sh, t1, t2):
    seed, w, h, kind = sh
    img = _img(seed, w, h, kind)
    low, high = min(t1, t2), max(t1, t2)
    stg = oracle.canny_r(img, low, high, stages=True)
    thr, edges = stg["thresh"], stg["edges"]
    assert set(np.unique(thr)) <= {0, 128, 255} and set(np.unique(edges)) <= {0, 255}
    # strong pixels survive, candidates may, nothing else appears (cannyEdgeD.cu:333-395)
    assert ((thr == 255) <= (edges == 255)).all() and ((edges == 255) <= (thr >= 128)).all()
    # hysteresis is idempotent: its output, fed back as a map of strong pixels, is a fixpoint
    assert np.array_equal(oracle.hysteresis(edges), edges)
    # raising the low threshold can only remove edges; raising the high one likewise
    if low < high:
        assert ((oracle.canny_r(img, low + 1, high) == 255) <= (edges == 255)).all()
    if high < 255:
        assert ((oracle.canny_r(img, low, high + 1) == 255) <= (edges == 255)).all()


@settings(max_examples=25, **SET)
@given(st.integers(0, 255), st.integers(1, 80), st.integers(1, 60))
def test_oracle_flat_frames(oracle, v, w, h):
    """A constant frame has a constant blur away from the border, so every edge lies in the ring of width 4 that the
    zero padding of the blur / Sobel / NMS stages disturbs (cannyEdgeD.cu:91-98, 142-149, 222-229); a zero frame has none.
    (Flips of the frame are NOT symmetries of the reference: the 25-term float chain and the closed / open ends of the
    direction bins depend on the order of rows and columns.)"""
    e = oracle.canny_r(np.full((h, w), v, np.uint8), 10, 40)
    inner = e[4:-4, 4:-4] if h > 8 and w > 8 else np.zeros((0, 0), np.uint8)
    assert not inner.any()
    if v == 0:
        assert not e.any()


@settings(max_examples=25, **SET)
@given(shape)
def test_oracle_mode_o_laws(oracle, sh):
    seed, w, h, kind = sh
    img = _img(seed, w, h, kind)
    e, pre = oracle.canny_o_stages(img, 50, 150)
    assert ((pre == 255) <= (e == 255)).all() and ((e == 255) <= (pre >= 128)).all()
    assert ((oracle.canny_o(img, 60, 150) == 255) <= (e == 255)).all()
    three = np.stack([img, img, img], axis=-1)   # equal channels: the first wins every tie
    assert np.array_equal(oracle.canny_o(three, 50, 150), e)


# ---- the HIP path ----------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@settings(max_examples=30, **SET)
@given(st.tuples(st.integers(0, 10_000), st.integers(1, 700), st.integers(1, 200), st.integers(0, 2)), st.integers(0, 255), st.integers(0, 255), st.booleans())
def test_gpu_matches_oracle_and_obeys_the_laws(oracle, sh, t1, t2, pipeline):
    import torch
    from cudacam_amd import api
    seed, w, h, kind = sh
    img = _img(seed, w, h, kind)
    low, high = min(t1, t2), max(t1, t2)
    with api.Context(w, h, 1, 1) as ctx:
        ctx.set_thresholds(low, high)
        ctx.set_option(api.OPT_PIPELINE, int(pipeline))
        ctx.set_option(api.OPT_DEBUG_TAPS, 1)
        edges = ctx.process(img)[0]
        thr = ctx.debug_tap(api.TAP_THRESH)[0]
        assert np.array_equal(edges, oracle.canny_r(img, low, high))
        assert ((thr == 255) <= (edges == 255)).all() and ((edges == 255) <= (thr >= 128)).all()
        # idempotence through the device hysteresis entry point: the map is its own fixpoint
        d_in = torch.from_numpy(edges).cuda()
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        pitch = w
        if w % 4 == 0:
            ctx.hysteresis_device(d_in.data_ptr(), pitch, pitch * h, d_out.data_ptr(), pitch, pitch * h, 1)
            ctx.sync()
            assert np.array_equal(d_out.cpu().numpy(), edges)


@pytest.mark.gpu
def test_gpu_full_size_laws(oracle):
    """BASELINE configs[1] at full size (1920x1080, a batch): threshold monotonicity and hysteresis idempotence on the
    whole batch, the oracle on two frames of it."""
    import torch
    from cudacam_amd import api
    frames = np.stack([synth.natural(1920, 1080, 300 + f) for f in range(6)])
    maps = {}
    with api.Context(1920, 1080, 1, 6) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        for low, high in ((10, 40), (11, 40), (10, 41), (40, 40)):
            ctx.set_thresholds(low, high)
            maps[(low, high)] = ctx.process(frames).copy()
        base = maps[(10, 40)]
        for other in ((11, 40), (10, 41), (40, 40)):
            assert ((maps[other] == 255) <= (base == 255)).all(), other
        for f in (0, 5):
            assert np.array_equal(base[f], oracle.canny_r(frames[f], 10, 40))
        d_in = torch.from_numpy(base).cuda()
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        ctx.hysteresis_device(d_in.data_ptr(), 1920, 1920 * 1080, d_out.data_ptr(), 1920, 1920 * 1080, 6)
        ctx.sync()
        assert np.array_equal(d_out.cpu().numpy(), base)
