"""In-process multi-device dispatch (cudacam_amd/shard.py run_on_devices): N frames on one context versus the same
frames cut into contiguous blocks over several contexts on several host threads -- byte-identical maps, in frame order.
With one visible GPU the "devices" are several contexts on device 0 (what the GPU test box has); with two or more
visible GPUs the same test also runs across real devices."""
import numpy as np
import pytest

from cudacam_amd import api, shard, synth

pytestmark = pytest.mark.gpu


def _ndev():
    import torch
    return torch.cuda.device_count()


def _frames(n, w=500, h=260):
    return np.stack([synth.natural(w, h, 900 + f) if f % 4 else synth.noise(w, h, 900 + f) for f in range(n)])


@pytest.mark.parametrize("world", [2, 3])
def test_blocks_over_contexts_match_single_context(oracle, world):
    frames = _frames(11)
    single = shard.run_on_devices(frames, [0])
    many = shard.run_on_devices(frames, [0] * world, batch=3, options=[(api.OPT_PIPELINE, 1)])
    assert np.array_equal(single, many)
    for f in (0, 5, 10):
        assert np.array_equal(single[f], oracle.canny_r(frames[f], 10, 40))


def test_blocks_over_real_devices_match_single_device(oracle):
    n = _ndev()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the cross-device run needs two (the same dispatch code ran on contexts of device 0 above)")
    frames = _frames(16)
    single = shard.run_on_devices(frames, [0])
    many = shard.run_on_devices(frames, list(range(min(n, 8))), batch=2)
    assert np.array_equal(single, many)


def test_three_channel_blocks(oracle):
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (5, 120, 336, 3), dtype=np.uint8)
    single = shard.run_on_devices(frames, [0])
    many = shard.run_on_devices(frames, [0, 0], batch=2)
    assert np.array_equal(single, many)
    assert np.array_equal(single[3], oracle.canny_r(frames[3], 10, 40))


def test_per_channel_blocks(oracle):
    """OPT_PER_CHANNEL returns three maps per input frame -- the dispatcher sizes its result for that (frame
    f's maps are rows 3f .. 3f+2) and refuses the option on one-channel frames."""
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, (5, 96, 336, 3), dtype=np.uint8)
    single = shard.run_on_devices(frames, [0], options=[(api.OPT_PER_CHANNEL, 1)])
    many = shard.run_on_devices(frames, [0, 0], batch=2, options=[(api.OPT_PER_CHANNEL, 1)])
    assert single.shape == (15, 96, 336)
    assert np.array_equal(single, many)
    for f, ch in ((0, 0), (3, 1), (4, 2)):
        assert np.array_equal(single[3 * f + ch], oracle.canny_r(np.ascontiguousarray(frames[f, :, :, ch]), 10, 40))
    with pytest.raises(ValueError):
        shard.run_on_devices(frames[..., 0], [0], options=[(api.OPT_PER_CHANNEL, 1)])
