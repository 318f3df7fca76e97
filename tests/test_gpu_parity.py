"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C ABI,
against the CPU oracle -- bit-exact, every stage and the final edge map."""
import ctypes as C
import os

import numpy as np
import pytest

from cudacam_amd import api, synth

pytestmark = pytest.mark.gpu

STAGE_KEYS = {
    api.CannyStage.MONO: "mono", api.CannyStage.GAUSSIAN: "blur", api.CannyStage.GRADIENT: "grad_disp",
    api.CannyStage.NMS: "nms", api.CannyStage.THRESH: "thresh", api.CannyStage.HYSTER: "edges",
}


def _diff(a, b, what):
    if np.array_equal(a, b):
        return
    bad = np.argwhere(a != b)
    first = [(tuple(int(v) for v in p), int(a[tuple(p)]), int(b[tuple(p)])) for p in bad[:8]]
    raise AssertionError(f"{what}: {len(bad)} of {a.size} differ; first (pos, hip, oracle): {first}")


def _images():
    yield "natural_640x480", synth.natural(640, 480, 1)
    yield "noise_641x479", synth.noise(641, 479, 2)
    yield "natural_31x33", synth.natural(31, 33, 3)
    yield "noise_5x5", synth.noise(5, 5, 4)
    yield "one_px", np.array([[200]], np.uint8)
    yield "flat100_40x36", synth.flat(40, 36, 100)
    yield "flat255_300x70", synth.flat(300, 70, 255)
    yield "zeros_64x64", synth.flat(64, 64, 0)
    yield "step255_v", synth.steps(260, 64, 255, "vertical")
    yield "step240_h", synth.steps(100, 90, 240, "horizontal")
    yield "step234_d", synth.steps(250, 250, 234, "diagonal")
    yield "serpentine_500x300", synth.serpentine(500, 300)
    yield "natural_249x17", synth.natural(249, 17, 5)
    yield "natural_248x65", synth.natural(248, 65, 6)
    yield "noise_497x130", synth.noise(497, 130, 7)


def test_selftest():
    api.selftest(0)


@pytest.mark.parametrize("name,img", list(_images()), ids=[n for n, _ in _images()])
def test_all_stages_mono(oracle, name, img):
    h, w = img.shape
    want = oracle.canny_r(img, 10, 40, stages=True)
    with api.Context(w, h, 1, 1) as ctx:
        for stage, key in STAGE_KEYS.items():
            got = ctx.process(img, stage)[0]
            _diff(got, want[key], f"{name} stage {stage.name}")


@pytest.mark.parametrize("split", [2, 1, 0])
@pytest.mark.parametrize("chunk", [7, 20, 44, 116, 1080])
def test_chunk_invariance(oracle, chunk, split):
    """Rows per work item (and the one-kernel / two-kernel form of the front path) never change the result."""
    img = synth.natural(700, 333, 11)
    want = oracle.canny_r(img, 10, 40)
    with api.Context(700, 333, 1, 1, front_split=split) as ctx:
        ctx.set_tuning(chunk, 4)
        _diff(ctx.process(img)[0], want, f"chunk {chunk} split {split}")


def _nms_queue_patterns():
    """Frames that steer k_nms through each of its paths: a few candidate lanes per wave-row (queued), many (wave-wide),
    candidates in the first / last lanes of a strip and in both lanes of an output byte, batches that fill exactly and
    entries left over at the end of a run."""
    rng = np.random.default_rng(11)
    w, h = 1000, 150                                     # 5 strips of 248 columns, the last one ragged
    a = np.full((h, w), 60, np.uint8)
    for x in (3, 247, 248, 249, 251, 252, 495, 496, 500, 743, 744, 991, 999):   # thin vertical lines at strip borders and lane pairs
        a[:, x] = 200
    yield "vertical_lines", a
    b = np.full((h, w), 60, np.uint8)
    b[40:43, :] = 220                                    # horizontal edges: every lane of the row has candidates
    b[90, 100:140] = 220                                 # a short one: about ten lanes
    b[120, 300:368] = 220                                # 17 lanes: just above the wave-wide threshold in one strip, below in the next
    yield "horizontal_lines", b
    c = np.full((h, w), 30, np.uint8)
    ys = rng.integers(3, h - 3, 400); xs = rng.integers(3, w - 3, 400)
    c[ys, xs] = 255                                      # isolated dots: one or two lanes per row, many rows
    yield "dots", c
    d = np.full((67, 300), 20, np.uint8)                 # one run that ends with a partly filled queue
    d[5:60, 150] = 250
    yield "short_run", d
    e = np.full((h, w), 90, np.uint8)
    for k in range(-h, w, 37):                           # diagonals: a candidate lane that moves by one column per row
        for y in range(h):
            if 0 <= k + y < w:
                e[y, k + y] = 230
    yield "diagonals", e


@pytest.mark.parametrize("name,img", list(_nms_queue_patterns()), ids=[n for n, _ in _nms_queue_patterns()])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_nms_queue_paths(oracle, name, img, pipeline):
    h, w = img.shape
    for low, high in ((10, 40), (60, 200)):
        want = oracle.canny_r(img, low, high)
        with api.Context(w, h, 1, 2) as ctx:
            ctx.set_thresholds(low, high)
            ctx.set_option(api.OPT_PIPELINE, pipeline)
            got = ctx.process(np.stack([img, img[::-1].copy()]), api.CannyStage.HYSTER)
            _diff(got[0], want, f"{name} {low}/{high} pipeline={pipeline}")
            _diff(got[1], oracle.canny_r(img[::-1].copy(), low, high), f"{name} flipped {low}/{high} pipeline={pipeline}")


@pytest.mark.parametrize("kind", ["noise", "flat", "steps", "natural"])
def test_fused_front_kernel(oracle, kind):
    """HC_OPT_FRONT_SPLIT = 0: the single fused kernel (no blur plane) gives the same maps, BGR included."""
    img = {"noise": lambda: synth.noise(517, 203, 3), "flat": lambda: synth.flat(517, 203, 100),
           "steps": lambda: synth.steps(517, 203, 90), "natural": lambda: synth.natural(517, 203, 3)}[kind]()
    want = oracle.canny_r(img, 10, 40)
    with api.Context(517, 203, 1, 1, front_split=0) as ctx:
        _diff(ctx.process(img)[0], want, f"fused {kind}")
    rng = np.random.default_rng(17)
    bgr = rng.integers(0, 256, (90, 260, 3), dtype=np.uint8)
    with api.Context(260, 90, 3, 1, front_split=0) as ctx:
        _diff(ctx.process(bgr)[0], oracle.canny_r(bgr, 10, 40), "fused bgr")


def test_bgr_input(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (120, 333, 3), dtype=np.uint8)
    want = oracle.canny_r(img, 10, 40, stages=True)
    with api.Context(333, 120, 3, 1) as ctx:
        for stage, key in STAGE_KEYS.items():
            _diff(ctx.process(img, stage)[0], want[key], f"bgr stage {stage.name}")


def test_bgr_fused_and_fallback(oracle):
    """BGR input: stage 0 is fused into the front kernel's load when rows hold whole 4-pixel groups
    (internal buffers always do); a tight caller pitch with W % 4 != 0 takes the k_gray path."""
    import torch
    rng = np.random.default_rng(11)
    for (w, h) in ((1920, 64), (641, 33), (250, 40)):
        img = np.ascontiguousarray(rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8))
        img[0, :, :, :] = np.repeat(synth.natural(w, h, 5)[:, :, None], 3, axis=2)  # grey-ish frame with structure
        img[0, :, :, 1] = np.clip(img[0, :, :, 1].astype(int) + 20, 0, 255)
        want = [oracle.canny_r(img[f], 10, 40) for f in range(2)]
        with api.Context(w, h, 3, 2) as ctx:
            got = ctx.process(img)                       # hc_upload path (internal pitched buffer): fused
            for f in range(2):
                _diff(got[f], want[f], f"bgr fused {w}x{h} frame {f}")
            d_in = torch.from_numpy(img).cuda()          # tight pitch 3*w
            d_out = torch.zeros((2, h, w), dtype=torch.uint8, device="cuda")
            ctx.run_device(d_in.data_ptr(), 3 * w, 3 * w * h, d_out.data_ptr(), w, w * h, 2)
            ctx.sync()
            for f in range(2):
                _diff(d_out[f].cpu().numpy(), want[f], f"bgr run_device {w}x{h} frame {f}")


def test_per_channel_mode(oracle):
    """HC_OPT_PER_CHANNEL: three edge maps per interleaved 3-channel frame (BASELINE config C5 at a
    small size), each equal to the detector run on that channel alone."""
    rng = np.random.default_rng(21)
    w, h = 501, 77
    img = np.stack([np.stack([synth.natural(w, h, 60 + 3 * f + c) for c in range(3)], axis=2) for f in range(2)])
    img[1, 10:40, 100:300, 2] = rng.integers(0, 256, (30, 200), dtype=np.uint8)
    with api.Context(w, h, 3, 2) as ctx:
        ctx.set_option(api.OPT_PER_CHANNEL, 1)
        ctx.upload(img)
        ctx.run(api.CannyStage.HYSTER, 2)
        got = ctx.download(6)
        for f in range(2):
            for c in range(3):
                _diff(got[3 * f + c], oracle.canny_r(np.ascontiguousarray(img[f, :, :, c]), 10, 40), f"per-channel frame {f} ch {c}")
        ctx.set_option(api.OPT_PER_CHANNEL, 0)
        _diff(ctx.process(img)[1], oracle.canny_r(img[1], 10, 40), "grey mode after per-channel mode")


@pytest.mark.parametrize("low,high", [(0, 0), (0, 255), (255, 255), (5, 250), (40, 10), (100, 101)])
def test_thresholds(oracle, low, high):
    img = synth.steps(300, 200, 250, "diagonal")
    img[50:150, 40:140] = synth.natural(100, 100, 9)
    with api.Context(300, 200, 1, 1) as ctx:
        ctx.set_thresholds(low, high)
        lo, hi = ctx.get_thresholds()
        assert (lo, hi) == (min(low, high), max(low, high))
        want = oracle.canny_r(img, lo, hi, stages=True)
        _diff(ctx.process(img, api.CannyStage.THRESH)[0], want["thresh"], "thresh")
        _diff(ctx.process(img, api.CannyStage.HYSTER)[0], want["edges"], "edges")


@pytest.mark.parametrize("w,h", [(3840, 96), (4097, 70), (7680, 48), (8184, 33), (2049, 40)])
def test_wide_frames(oracle, w, h):
    """4K / 8K row widths: the bit-plane rows span 2 or 4 dwords per lane in the hysteresis kernel."""
    img = synth.natural(w, h, 300 + w)
    img[:, w // 2 - 700: w // 2 + 700] = synth.serpentine(1400, h, pitch=10, margin=3)
    want = oracle.canny_r(img, 10, 40, stages=True)
    with api.Context(w, h, 1, 1) as ctx:
        _diff(ctx.process(img, api.CannyStage.THRESH)[0], want["thresh"], f"{w}x{h} thresh")
        _diff(ctx.process(img, api.CannyStage.HYSTER)[0], want["edges"], f"{w}x{h} edges")


def test_4k_frame(oracle):
    img = synth.natural(3840, 2160, 4242)
    want = oracle.canny_r(img, 10, 40)
    with api.Context(3840, 2160, 1, 2) as ctx:
        got = ctx.process(np.stack([img, img[::-1].copy()]))
        _diff(got[0], want, "4K frame 0")
        _diff(got[1], oracle.canny_r(img[::-1].copy(), 10, 40), "4K frame 1")


def test_saturate_option(oracle):
    img = synth.steps(300, 200, 255, "diagonal")
    img[20:120, 150:290] = synth.noise(140, 100, 5)
    with api.Context(300, 200, 1, 1) as ctx:
        ctx.set_option(api.OPT_NMS_SATURATE, 1)
        for (lo, hi) in ((10, 40), (200, 255), (255, 255)):
            ctx.set_thresholds(lo, hi)
            want = oracle.canny_r(img, lo, hi, stages=True, saturate=True)
            for stage in (api.CannyStage.NMS, api.CannyStage.THRESH, api.CannyStage.HYSTER):
                _diff(ctx.process(img, stage)[0], want[STAGE_KEYS[stage]], f"saturate {lo}/{hi} {stage.name}")
        assert (oracle.canny_r(img, 10, 40, saturate=True) != oracle.canny_r(img, 10, 40)).any()


def test_batch_1080p(oracle):
    frames = np.stack([synth.natural(1920, 1080, 100 + i) for i in range(3)] + [synth.noise(1920, 1080, 200)])
    want = oracle.canny_r_batch(frames, 10, 40, threads=8)
    with api.Context(1920, 1080, 1, 4) as ctx:
        got = ctx.process(frames)
        for f in range(4):
            _diff(got[f], want[f], f"1080p frame {f}")
        assert ctx.hysteresis_info()[0] >= 1


def test_hysteresis_device_adversarial(oracle):
    import torch
    cases = {
        "serpentine_300x200": synth.thresh_map_serpentine(300, 200),
        "serpentine_1000x1100": synth.thresh_map_serpentine(1000, 1100),
        "random_dense": synth.thresh_map_random(777, 555, 3, 0.45, 0.002),
        "random_sparse": synth.thresh_map_random(1920, 1080, 4, 0.30, 0.0005),
        "all_candidates_one_seed": np.full((130, 520), 128, np.uint8),
        # frames wider than one 2048-column panel: the path crosses the panel seams (and the row tiles) again and again
        "serpentine_4500x300": synth.thresh_map_serpentine(4500, 300),
        "serpentine_8184x70": synth.thresh_map_serpentine(8184, 70),
        "random_wide": synth.thresh_map_random(6100, 300, 9, 0.45, 0.001),
        "all_candidates_one_seed_wide": np.full((300, 5000), 128, np.uint8),
        "diagonal_across_seams": np.zeros((600, 4200), np.uint8),
    }
    cases["all_candidates_one_seed"][129, 519] = 255
    cases["all_candidates_one_seed_wide"][299, 4999] = 255
    dg = cases["diagonal_across_seams"]   # a 1-px anti-diagonal and a diagonal through the tile corners at (256, 2048)
    for i in range(600):
        dg[i, 2048 - 256 + i] = 128
        dg[i, 2048 + 255 - i] = 128
    dg[0, 2048 - 256] = 255
    dg[599, 2048 + 255 - 599] = 255
    for name, t in cases.items():
        h, w = t.shape
        want = oracle.hysteresis(t)
        pitch = (w + 3) // 4 * 4
        buf = np.zeros((h, pitch), np.uint8)
        buf[:, :w] = t
        d_in = torch.from_numpy(buf).cuda()
        d_out = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
        for launches in (6, 1):
            with api.Context(w, h, 1, 1) as ctx:
                ctx.set_tuning(0, launches)
                ctx.hysteresis_device(d_in.data_ptr(), pitch, pitch * h, d_out.data_ptr(), pitch, pitch * h, 1)
                ctx.sync()
                got = d_out.cpu().numpy()[:, :w]
                _diff(got, want, f"{name} ({launches} queued launches)")
                work, cont = ctx.hysteresis_info()
                assert work >= 1
                if name.startswith("serpentine_1000") and launches == 1:
                    assert cont == 1  # the path crosses tile boundaries: one queued launch cannot finish it


def test_run_device_unaligned_and_torch_stream(oracle):
    import torch
    img = synth.natural(641, 479, 21)
    want = oracle.canny_r(img, 10, 40)
    d_in = torch.from_numpy(img).cuda()           # tight pitch 641: not a multiple of 4
    d_out = torch.zeros_like(d_in)
    with api.Context(641, 479, 1, 1) as ctx:
        # torch's default current stream is the null stream (handle 0): the run is queued on that very stream, behind the
        # kernels that produced d_in / zeroed d_out
        assert torch.cuda.current_stream().cuda_stream == 0
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.run_device(d_in.data_ptr(), 641, 641 * 479, d_out.data_ptr(), 641, 641 * 479, 1)
        ctx.sync()
        _diff(d_out.cpu().numpy(), want, "unaligned run_device")
        assert ctx.last_run_info() == (True, True, 2)   # both buffers went through the internal pitched ones: reported, not hidden
        # a side stream of the caller: producer and detector on it, no host synchronisation in between
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            big = torch.from_numpy(np.ascontiguousarray(img[::-1])).cuda(non_blocking=True)
            d_in2 = big.clone()
            d_out2 = torch.full_like(d_in2, 7)
            ctx.set_stream(side.cuda_stream)
            ctx.run_device(d_in2.data_ptr(), 641, 641 * 479, d_out2.data_ptr(), 641, 641 * 479, 1)
        ctx.sync()
        _diff(d_out2.cpu().numpy(), oracle.canny_r(np.ascontiguousarray(img[::-1]), 10, 40), "run_device on a side stream")
        ctx.use_own_stream()
        torch.cuda.synchronize()
        ctx.run_device(d_in.data_ptr(), 641, 641 * 479, d_out.data_ptr(), 641, 641 * 479, 1)
        ctx.sync()
        _diff(d_out.cpu().numpy(), want, "own stream again")


MODE_O_IMAGES = [
    ("natural_640x480", lambda: synth.natural(640, 480, 1), 50, 150),
    ("noise_641x479", lambda: synth.noise(641, 479, 2), 100, 300),
    ("noise_low_thresholds", lambda: synth.noise(333, 222, 5), 0, 40),
    ("natural_31x33", lambda: synth.natural(31, 33, 3), 20, 60),
    ("one_px", lambda: np.array([[200]], np.uint8), 50, 150),
    ("five", lambda: synth.noise(5, 5, 4), 10, 30),
    ("flat255", lambda: synth.flat(300, 70, 255), 50, 150),
    ("step_v", lambda: synth.steps(260, 64, 255, "vertical"), 50, 150),
    ("step_d", lambda: synth.steps(250, 250, 120, "diagonal"), 50, 150),
    ("serpentine", lambda: synth.serpentine(500, 300, amp=30, seed_amp=200), 50, 150),
    ("natural_1080p", lambda: synth.natural(1920, 1080, 9), 50, 150),
    ("natural_4k_strip", lambda: synth.natural(3840, 100, 10), 30, 90),
    ("swapped_thresholds", lambda: synth.natural(200, 100, 8), 150, 50),
]


@pytest.mark.parametrize("name,make,low,high", MODE_O_IMAGES, ids=[m[0] for m in MODE_O_IMAGES])
def test_mode_o_matches_cv_canny_restatement(oracle, name, make, low, high):
    """Mode O: the HIP path against the oracle's restatement of cv::Canny(img, low, high, 3, false)
    (OpenCV itself is not installed anywhere in this pipeline: parity with the real library is unpinned)."""
    img = make()
    h, w = img.shape
    want = oracle.canny_o(img, low, high)
    with api.Context(w, h, 1, 1, api.MODE_O) as ctx:
        ctx.set_thresholds(low, high)
        _diff(ctx.process(img)[0], want, f"mode O {name}")


@pytest.mark.parametrize("name,make,low,high", [m for m in MODE_O_IMAGES if m[0] in (
    "natural_640x480", "noise_641x479", "noise_low_thresholds", "five", "flat255", "step_d", "serpentine", "natural_4k_strip")],
    ids=lambda v: v if isinstance(v, str) else None)
def test_mode_o_l2gradient(oracle, name, make, low, high):
    """HC_OPT_L2_GRADIENT: cv::Canny(img, low, high, 3, true) -- squared magnitude against squared thresholds."""
    img = make()
    h, w = img.shape
    want = oracle.canny_o(img, low, high, l2gradient=True)
    with api.Context(w, h, 1, 1, api.MODE_O) as ctx:
        ctx.set_thresholds(low, high)
        ctx.set_option(api.OPT_L2_GRADIENT, 1)
        _diff(ctx.process(img)[0], want, f"mode O L2 {name}")


@pytest.mark.parametrize("w,h,mode", [(500, 300, "R"), (641, 203, "R"), (4100, 150, "R"), (2052, 270, "R"), (500, 300, "O"), (4100, 150, "O")])
def test_pipelined_runs(oracle, w, h, mode):
    """HC_OPT_PIPELINE: back-to-back device runs overlap (front of run i+1 / hysteresis of run i);
    every run's output must still be exactly the oracle's after hc_sync.  Widths that are multiples of 4 take the
    provisional-map path (k_nms / k_front_o write the strong pixels, the hysteresis patches 16-pixel groups), 641 does
    not; 2052 and 4100 span several hysteresis panels."""
    import torch
    nb = 3
    batches = [np.stack([synth.natural(w, h, 40 + 10 * r + f) for f in range(nb)]) for r in range(5)]
    batches[3] = np.stack([synth.serpentine(w, h, amp=30 if mode == "O" else 20, seed_amp=200 if mode == "O" else 120) for f in range(nb)])  # long thin chains: many patches
    if mode == "R":
        want = [oracle.canny_r_batch(b, 10, 40, threads=4) for b in batches]
    else:
        want = [oracle.canny_o_batch(b, 50, 150, threads=4) for b in batches]
    pitch = (w + 3) // 4 * 4
    d_in = []
    for b in batches:
        buf = np.zeros((nb, h, pitch), np.uint8)
        buf[:, :, :w] = b
        d_in.append(torch.from_numpy(buf).cuda())
    d_out = [torch.zeros_like(t) for t in d_in]
    with api.Context(w, h, 1, nb, api.MODE_R if mode == "R" else api.MODE_O) as ctx:
        if mode == "O":
            ctx.set_thresholds(50, 150)
        ctx.set_option(api.OPT_PIPELINE, 1)
        for rep in range(2):
            for r in range(5):
                ctx.run_device(d_in[r].data_ptr(), pitch, pitch * h, d_out[r].data_ptr(), pitch, pitch * h, nb)
            ctx.sync()
            for r in range(5):
                got = d_out[r].cpu().numpy()[:, :, :w]
                for f in range(nb):
                    _diff(got[f], want[r][f], f"pipelined rep {rep} run {r} frame {f}")
                d_out[r].fill_(rep + 7)   # stale bytes must not survive: the next pass rewrites (or patches) everything
        ctx.set_option(api.OPT_PIPELINE, 0)
        ctx.run_device(d_in[0].data_ptr(), pitch, pitch * h, d_out[0].data_ptr(), pitch, pitch * h, nb)
        ctx.sync()
        _diff(d_out[0].cpu().numpy()[1, :, :w], want[0][1], "plain mode after pipelined mode")


@pytest.mark.parametrize("w", [4500, 900])
@pytest.mark.parametrize("grid", ["1", "3"])
def test_hysteresis_worklists_with_tiny_grids(oracle, grid, w):
    """The worklist form of the hysteresis (frames wider than one 2048-column panel; one-panel streams that need 20
    launches or more): launches >= 1 take their tiles from lists, one entry per workgroup; entries beyond the grid are
    handed on to the next launch.  HC_OPT_TEST_HYST_LATE_GRID forces the lists and grids of 1 and 3
    workgroups, so nearly every entry of every launch takes that road -- serpentine chains that cross panel seams and
    row tiles, plain and pipelined, must still reach the exact fixpoint (through the host-side continuation when the
    queued launches run out)."""
    import torch
    h, nb = 200, 2
    frames = np.stack([synth.serpentine(w, h, amp=20, seed_amp=120), synth.natural(w, h, 77)])
    want = oracle.canny_r_batch(frames, 10, 40, threads=4)
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_TEST_HYST_LATE_GRID, int(grid))
        got = ctx.process(frames)
        for f in range(nb):
            _diff(got[f], want[f], f"tiny late grid {grid}, frame {f}")
        pitch = (w + 7) // 8 * 8
        d_in = torch.from_numpy(np.ascontiguousarray(np.pad(frames, ((0, 0), (0, 0), (0, pitch - w))))).cuda()
        d_out = [torch.zeros_like(d_in) for _ in range(2)]
        ctx.set_option(api.OPT_PIPELINE, 1)
        for r in range(4):
            ctx.run_device(d_in.data_ptr(), pitch, pitch * h, d_out[r % 2].data_ptr(), pitch, pitch * h, nb)
        ctx.sync()
        for o in d_out:
            for f in range(nb):
                _diff(o.cpu().numpy()[f, :, :w], want[f], f"tiny late grid {grid}, pipelined, frame {f}")


@pytest.mark.parametrize("nbuf", [1, 2, 3, 4, 5])
def test_small_batches_keep_four_runs_in_flight(oracle, nbuf):
    """Pipelined runs of small batches rotate through four sets of bit planes, each hysteresis chain on a stream of its
    own.  Whatever the number of output buffers the caller cycles through -- a run into memory that an older run still
    in flight writes must wait for it -- every buffer ends up holding the map of the last run that wrote it; then the
    same context takes a big batch (back to the two-slot ring) and small ones again."""
    import torch
    w, h, nb = 640, 200, 2
    runs = [np.stack([synth.natural(w, h, 300 + 7 * r + f) if (r + f) % 3 else synth.serpentine(w, h, amp=20, seed_amp=120) for f in range(nb)]) for r in range(11)]
    want = [oracle.canny_r_batch(b, 10, 40, threads=4) for b in runs]
    d_in = [torch.from_numpy(b).cuda() for b in runs]
    d_out = [torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    big = np.tile(np.stack([synth.natural(w, h, 900 + f) for f in range(40)]), (100, 1, 1))   # 4000 frames = 0.51 G pixels: the two-slot ring
    with api.Context(w, h, 1, 4000) as ctx:
        assert ctx.pipeline_depth(nb) == 1
        ctx.set_option(api.OPT_PIPELINE, 1)
        assert ctx.pipeline_depth(nb) == 4 and ctx.pipeline_depth(1400) == 4 and ctx.pipeline_depth(4000) == 3   # 0.5 G pixels = 3906 of these frames
        for rep in range(2):
            last = {}
            for r in range(len(runs)):
                ctx.run_device(d_in[r].data_ptr(), w, w * h, d_out[r % nbuf].data_ptr(), w, w * h, nb)
                last[r % nbuf] = r
            ctx.sync()
            for b, r in last.items():
                got = d_out[b].cpu().numpy()
                for f in range(nb):
                    _diff(got[f], want[r][f], f"{nbuf} buffers, rep {rep}, run {r}, frame {f}")
            if rep == 0:
                got = ctx.process(big)
                for f in (0, 1700, 3999):
                    _diff(got[f], oracle.canny_r(big[f], 10, 40), f"big batch between the small ones, frame {f}")


def test_big_batches_take_a_third_slot_when_the_hysteresis_chain_bounds_the_step(oracle):
    """Big pipelined batches rotate through two slots, and through three once the context has seen the hysteresis chain of
    a run end after the front kernel of the next one (hipcanny.hip, watch_chain).  Whether that happens depends on the
    content and on the machine, so the rule is also walked through its transitions by hand (HC_OPT_PIPELINE_SLOTS 20 / 21:
    every chain counts as the longer / the shorter): 2 -> 3 on trial after three runs, kept or given back after ten,
    3 -> 2 after sixteen, the ring resized with runs in flight -- and the maps are the oracle's whichever ring is in use, with three output buffers and with two (a
    run into memory that an older run still writes waits for it), and with the ring fixed at 3 or 2."""
    import torch
    w, h, nb = 640, 200, 4000   # 0.51 G pixels: a big batch
    uniq = np.stack([synth.serpentine(w, h, amp=20 + (f % 5), seed_amp=120) if f % 2 else synth.natural(w, h, 700 + f) for f in range(8)])
    want = oracle.canny_r_batch(uniq, 10, 40, threads=4)
    d_in = torch.from_numpy(np.tile(uniq, (nb // 8, 1, 1))).cuda()
    d_out = [torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda") for _ in range(3)]

    def runs(ctx, n, nbuf, tag):
        for r in range(n):
            ctx.run_device(d_in.data_ptr(), w, w * h, d_out[r % nbuf].data_ptr(), w, w * h, nb)
        ctx.sync()
        for k, o in enumerate(d_out[:nbuf]):
            got = o[:: nb // 16].cpu().numpy()   # frames 0, 250, 500, ...: every distinct frame twice
            for j in range(got.shape[0]):
                _diff(got[j], want[(j * (nb // 16)) % 8], f"{tag}, buffer {k}, frame {j * (nb // 16)}")
            o.zero_()

    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        assert ctx.pipeline_depth(nb) == 3
        runs(ctx, 8, 3, "automatic")
        assert ctx.pipeline_slots_in_use() in (2, 3)
        ctx.set_option(api.OPT_PIPELINE_SLOTS, 20)
        runs(ctx, 8, 3, "told that every chain outlasts the next front kernel: a third slot on trial")
        assert ctx.pipeline_slots_in_use() == 3
        runs(ctx, 8, 3, "the trial ends (ten runs), the slot is kept")
        assert ctx.pipeline_slots_in_use() == 3
        runs(ctx, 5, 2, "three slots, two output buffers")
        ctx.set_option(api.OPT_PIPELINE_SLOTS, 21)
        runs(ctx, 26, 3, "told that every chain ends first: back to two after sixteen runs")
        assert ctx.pipeline_slots_in_use() == 2
        ctx.set_option(api.OPT_PIPELINE_SLOTS, 20)
        runs(ctx, 16, 3, "and up again")
        assert ctx.pipeline_slots_in_use() == 3
        ctx.set_option(api.OPT_PIPELINE_SLOTS, 21)
        runs(ctx, 26, 3, "the way down takes twice as many runs the second time")
        assert ctx.pipeline_slots_in_use() == 3
        runs(ctx, 12, 3, "... and is taken")
        assert ctx.pipeline_slots_in_use() == 2
        ctx.set_option(api.OPT_PIPELINE_SLOTS, 20)
        runs(ctx, 8, 3, "a trial")
        assert ctx.pipeline_slots_in_use() == 3
        ctx.set_option(api.OPT_PIPELINE_SLOTS, 21)
        runs(ctx, 16, 3, "... that does not pay: the third slot is given back when it ends")
        assert ctx.pipeline_slots_in_use() == 2
        for forced in (2, 3):
            ctx.set_option(api.OPT_PIPELINE_SLOTS, forced)
            assert ctx.pipeline_depth(nb) == forced
            runs(ctx, 7, 3, f"ring fixed at {forced}")
            assert ctx.pipeline_slots_in_use() == forced
        with pytest.raises(api.HipCannyError):
            ctx.set_option(api.OPT_PIPELINE_SLOTS, 5)


def test_python_mirror_of_reference_operator(oracle):
    img = synth.natural(320, 200, 31)
    pipe = api.cvPipeline(0, 320, 200, 1)
    assert pipe.getLowThreshold() == 10 and pipe.getHighThreshold() == 40 and pipe.isCudaProfilingEnabled()
    assert pipe.process(img, api.CannyStage.HYSTER) is True
    _diff(pipe.output(), oracle.canny_r(img, 10, 40), "cvPipeline.process")
    assert pipe.process(np.zeros((0, 0), np.uint8), api.CannyStage.HYSTER) is False           # blank frame
    assert pipe.process(np.zeros((200, 320), np.float32), api.CannyStage.HYSTER) is False     # wrong type
    pipe.setLowThreshold(90)      # min(low, high) -> 40 (cannyEdgeH.hpp:25)
    assert pipe.getLowThreshold() == 40
    pipe.setHighThreshold(5)      # max(high, low) -> 40
    assert pipe.getHighThreshold() == 40
    assert api.TimerManager.Get().getAverageTime(api.CANNY_STAGES[api.CannyStage.HYSTER]) >= 0.0


def test_context_reuse_sequences(oracle):
    """One context, many different calls in a row: batch sizes going up and down, thresholds, options and stage taps
    changing between runs, pipelined and plain -- no state may leak from one run into the next."""
    import torch
    w, h, nb = 700, 300, 4
    frames = np.stack([synth.natural(w, h, 60 + f) for f in range(nb)])
    noise = np.stack([synth.noise(w, h, 80 + f) for f in range(nb)])
    with api.Context(w, h, 1, nb, front_split=1) as ctx:   # (the test library: the runs below switch between the round-1 kernels)
        for pipelined in (0, 1, 0):
            ctx.set_option(api.OPT_PIPELINE, pipelined)
            for (src, n, low, high, split, sat) in [(frames, 4, 10, 40, 1, 0), (noise, 1, 30, 90, 0, 1), (frames, 2, 0, 255, 1, 1),
                                                     (noise, 4, 5, 6, 1, 0), (frames, 3, 10, 40, 0, 0)]:
                ctx.set_thresholds(low, high)
                ctx.set_option(api.OPT_FRONT_SPLIT, split)
                ctx.set_option(api.OPT_NMS_SATURATE, sat)
                got = ctx.process(src[:n])
                for f in range(n):
                    _diff(got[f], oracle.canny_r(src[f], low, high, saturate=bool(sat)), f"pipelined {pipelined} n {n} thr {low}/{high} split {split} sat {sat} frame {f}")
            ctx.set_option(api.OPT_NMS_SATURATE, 0)
            ctx.set_thresholds(10, 40)
            st = oracle.canny_r(frames[1], 10, 40, stages=True)
            got = ctx.process(frames[1:2], api.CannyStage.GAUSSIAN)
            _diff(got[0], st["blur"], "gaussian tap after fused runs")
        # device path, pipelined, batches of different size back to back without a sync in between
        ctx.set_option(api.OPT_PIPELINE, 1)
        ctx.set_option(api.OPT_FRONT_SPLIT, 1)
        d_in = torch.from_numpy(np.concatenate([frames, noise])).cuda()
        d_out = torch.zeros_like(d_in)
        plan = [(0, 4), (4, 2), (6, 1), (7, 1), (1, 3)]
        for (f0, n) in plan:
            ctx.run_device(d_in[f0].data_ptr(), w, w * h, d_out[f0].data_ptr(), w, w * h, n)
        ctx.sync()
        allf = np.concatenate([frames, noise])
        got = d_out.cpu().numpy()
        for f in range(8):
            _diff(got[f], oracle.canny_r(allf[f], 10, 40), f"pipelined device runs, frame {f}")


def test_two_contexts_from_two_threads(oracle):
    """Contexts are independent (include/hipcanny.h): two of them, different sizes and modes, driven from two threads."""
    import threading
    jobs = [(synth.natural(900, 400, 5), "R", 10, 40), (synth.noise(640, 333, 6), "O", 60, 180)]
    want = [oracle.canny_r(jobs[0][0], 10, 40), oracle.canny_o(jobs[1][0], 60, 180)]
    errs = []

    def work(k):
        img, mode, lo, hi = jobs[k]
        try:
            with api.Context(img.shape[1], img.shape[0], 1, 2, api.MODE_R if mode == "R" else api.MODE_O) as ctx:
                ctx.set_thresholds(lo, hi)
                for _ in range(25):
                    got = ctx.process(np.stack([img, img]))
                    if not (np.array_equal(got[0], want[k]) and np.array_equal(got[1], want[k])):
                        errs.append(f"context {k}: mismatch")
                        return
        except Exception as e:  # noqa: BLE001
            errs.append(f"context {k}: {e!r}")

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("per_channel", [0, 1])
def test_pipelined_three_channel_runs(oracle, per_channel):
    """Pipelined device runs on interleaved 3-channel input: grey conversion fused into k_blur, or one map per channel
    (3 output frames per input frame; the provisional map is written per output frame)."""
    import torch
    w, h, nb = 520, 260, 2
    rng = np.random.default_rng(77)
    batches = [np.stack([np.stack([synth.natural(w, h, 300 + 7 * r + 3 * f + c) for c in range(3)], axis=-1) for f in range(nb)]) for r in range(4)]
    if per_channel:
        want = [np.stack([oracle.canny_r(np.ascontiguousarray(b[f, :, :, c]), 10, 40) for f in range(nb) for c in range(3)]) for b in batches]
    else:
        want = [np.stack([oracle.canny_r(b[f], 10, 40) for f in range(nb)]) for b in batches]
    d_in = [torch.from_numpy(b).cuda() for b in batches]
    n_out = nb * (3 if per_channel else 1)
    d_out = [torch.full((n_out, h, w), 9, dtype=torch.uint8, device="cuda") for _ in batches]
    with api.Context(w, h, 3, nb) as ctx:
        ctx.set_option(api.OPT_PER_CHANNEL, per_channel)
        ctx.set_option(api.OPT_PIPELINE, 1)
        for r in range(4):
            ctx.run_device(d_in[r].data_ptr(), 3 * w, 3 * w * h, d_out[r].data_ptr(), w, w * h, nb)
        ctx.sync()
        for r in range(4):
            got = d_out[r].cpu().numpy()
            for f in range(n_out):
                _diff(got[f], want[r][f], f"3-channel pipelined (per_channel {per_channel}) run {r} map {f}")


def test_pipelined_runs_into_one_output_buffer(oracle):
    """A caller that hands every pipelined run the same output buffer loses the earlier maps, but after hc_sync the
    buffer must hold the LAST run's map exactly: no late patch of an earlier run's hysteresis may survive in it
    (the library drops the provisional-map shortcut when consecutive outputs overlap)."""
    import torch
    w, h, nb = 600, 320, 2
    runs = [np.stack([synth.natural(w, h, 500 + 13 * r + f) for f in range(nb)]) for r in range(6)]
    runs[4] = np.stack([synth.serpentine(w, h) for _ in range(nb)])
    d_in = [torch.from_numpy(b).cuda() for b in runs]
    d_out = torch.zeros((nb, h, w), dtype=torch.uint8, device="cuda")
    with api.Context(w, h, 1, nb) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        for last in (5, 3):
            for r in range(last + 1):
                ctx.run_device(d_in[r].data_ptr(), w, w * h, d_out.data_ptr(), w, w * h, nb)
            ctx.sync()
            got = d_out.cpu().numpy()
            for f in range(nb):
                _diff(got[f], oracle.canny_r(runs[last][f], 10, 40), f"one output buffer, last run {last}, frame {f}")


def test_pipelined_one_buffer_with_host_continuation(oracle):
    """a run whose queued hysteresis launches do not reach the fixpoint is continued from the host when its
    slot is completed -- after the NEXT run may already have written the same output buffer.  With one queued launch
    and a serpentine frame the continuation is certain; the buffer must still hold the last run's map."""
    import torch
    w, h = 1000, 1100
    serp = synth.serpentine(w, h)
    nat = synth.natural(w, h, 77)
    d_serp, d_nat = torch.from_numpy(serp[None]).cuda(), torch.from_numpy(nat[None]).cuda()
    d_out = torch.zeros((1, h, w), dtype=torch.uint8, device="cuda")
    want_nat, want_serp = oracle.canny_r(nat, 10, 40), oracle.canny_r(serp, 10, 40)
    with api.Context(w, h, 1, 1) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        ctx.set_tuning(0, 1)
        for order, want in (((d_serp, d_nat), want_nat), ((d_nat, d_serp), want_serp), ((d_serp, d_serp, d_nat), want_nat)):
            for d in order:
                ctx.run_device(d.data_ptr(), w, w * h, d_out.data_ptr(), w, w * h, 1)
            ctx.sync()
            _diff(d_out.cpu().numpy()[0], want, f"one buffer, {len(order)} runs, continuation")


@pytest.mark.parametrize("k", [3, 4])
def test_pipelined_mixed_schedule_continuation(oracle, k):
    """a pipelined one-panel run queues launches 0-1 with a workgroup per tile, launch 2 writes the first
    worklist and the rest take lists.  With only 3 or 4 launches queued on a serpentine frame the host-side continuation
    is certain, and it must replay THAT schedule (Slot::mixed_from), not the parameters of the run's last launch."""
    import torch
    w, h = 1000, 2300
    # a weak vertical line down the whole frame with one strong head: every one of the 18 row tiles must be crossed, one
    # per launch (the boustrophedon frame of synth.serpentine settles in three launches at this size)
    serp = np.zeros((h, w), np.uint8)
    serp[:, 100:140] = 20
    for r in range(20):   # the head fades into the line, so that the strong edge and the weak one are connected
        serp[r, 100:140] = 120 - 5 * r
    nat = synth.natural(w, h, 78)
    frames = np.stack([serp, nat])
    want = oracle.canny_r_batch(frames, 10, 40, threads=4)
    d_in = torch.from_numpy(frames).cuda()
    d_out = [torch.zeros_like(d_in) for _ in range(2)]
    with api.Context(w, h, 1, 2) as ctx:
        ctx.set_option(api.OPT_PIPELINE, 1)
        ctx.set_tuning(0, k)
        ctx.hysteresis_totals(reset=True)
        for r in range(3):
            ctx.run_device(d_in.data_ptr(), w, w * h, d_out[r % 2].data_ptr(), w, w * h, 2)
        ctx.sync()
        runs, continued, work, queued = ctx.hysteresis_totals()
        for o in d_out:
            for f in range(2):
                _diff(o.cpu().numpy()[f], want[f], f"mixed schedule, {k} launches queued, frame {f}")
        assert runs == 3 and queued == 3 * k and continued >= 1, (runs, continued, work, queued)


def test_last_run_info_reports_form_and_staging(oracle):
    """Aligned caller buffers are used in place by k_front8; a row that does not hold whole 8-pixel groups is staged
    through the internal pitched buffer (and says so); mode O reports k_front8o (3), or the 4-px k_front_o (-1) when
    asked for a 4-px form."""
    import torch
    img = synth.natural(640, 100, 3)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    with api.Context(640, 100, 1, 1) as ctx:
        ctx.run_device(d_in.data_ptr(), 640, 640 * 100, d_out.data_ptr(), 640, 640 * 100, 1)
        ctx.sync()
        assert ctx.last_run_info() == (False, False, 2)
        _diff(d_out.cpu().numpy(), oracle.canny_r(img, 10, 40), "in place")
    img8 = synth.natural(800, 60, 5)
    d_in8 = torch.from_numpy(img8).cuda()
    d_out8 = torch.zeros_like(d_in8)
    torch.cuda.synchronize()
    with api.Context(800, 60, 1, 1) as ctx:
        ctx.run_device(d_in8.data_ptr(), 800, 800 * 60, d_out8.data_ptr(), 800, 800 * 60, 1)
        ctx.sync()
        assert ctx.last_run_info() == (False, False, 2)
        _diff(d_out8.cpu().numpy(), oracle.canny_r(img8, 10, 40), "in place, 800 columns")
    img2 = synth.natural(644, 60, 4)   # pitch 644: a multiple of 4 but not of 8
    d_in2 = torch.from_numpy(img2).cuda()
    d_out2 = torch.zeros_like(d_in2)
    torch.cuda.synchronize()
    with api.Context(644, 60, 1, 1) as ctx:
        ctx.run_device(d_in2.data_ptr(), 644, 644 * 60, d_out2.data_ptr(), 644, 644 * 60, 1)
        ctx.sync()
        assert ctx.last_run_info() == (True, False, 2)   # staged onto the 8-px kernel (round 2: fell back to k_blur + k_nms)
        _diff(d_out2.cpu().numpy(), oracle.canny_r(img2, 10, 40), "ragged tight rows, staged")
    with api.Context(640, 100, 1, 1, mode=api.MODE_O) as ctx:
        ctx.run_device(d_in.data_ptr(), 640, 640 * 100, d_out.data_ptr(), 640, 640 * 100, 1)
        ctx.sync()
        assert ctx.last_run_info() == (False, False, 3)
        _diff(d_out.cpu().numpy(), oracle.canny_o(img, 50, 150), "mode O in place")
        ctx.set_option(api.OPT_FRONT_SPLIT, 1)
        ctx.run_device(d_in.data_ptr(), 640, 640 * 100, d_out.data_ptr(), 640, 640 * 100, 1)
        ctx.sync()
        assert ctx.last_run_info() == (False, False, -1)
        _diff(d_out.cpu().numpy(), oracle.canny_o(img, 50, 150), "mode O, 4-px kernel")


def test_download_begin_end(oracle):
    """hc_download_begin queues the device -> host copy behind the run and returns; hc_download_end waits, verifies the
    convergence and -- when the hysteresis had to be continued from the host (one launch queued, a frame that needs 18) --
    repeats the copy: the host buffer must hold the final maps either way."""
    lib = api.load_library()
    w, h = 1000, 2300
    line = np.zeros((h, w), np.uint8)
    line[:, 100:140] = 20
    for r in range(20):
        line[r, 100:140] = 120 - 5 * r
    frames = np.stack([line, synth.natural(w, h, 79)])
    want = oracle.canny_r_batch(frames, 10, 40, threads=4)
    hout = lib.hc_host_alloc(2 * w * h)
    try:
        for launches in (0, 1):
            with api.Context(w, h, 1, 2) as ctx:
                ctx.set_tuning(0, launches)
                ctx.hysteresis_totals(reset=True)
                for _ in range(2):
                    ctx.upload(frames)
                    ctx.run(api.CannyStage.HYSTER, 2)
                    api._ck(lib.hc_download_begin(ctx.handle, C.c_void_p(hout), w, w * h, 2))
                    assert lib.hc_download_begin(ctx.handle, C.c_void_p(hout), w, w * h, 2) != 0   # one download at a time
                    api._ck(lib.hc_download_end(ctx.handle))
                    got = np.ctypeslib.as_array((C.c_uint8 * (2 * w * h)).from_address(hout)).reshape(2, h, w)
                    for f in range(2):
                        _diff(got[f], want[f], f"download_begin/end, {launches} launches queued, frame {f}")
                    got[:] = 7
                assert lib.hc_download_end(ctx.handle) != 0   # nothing in flight
                if launches == 1:
                    assert ctx.hysteresis_totals()[1] == 2, "the one-launch runs were not continued from the host"
    finally:
        lib.hc_host_free(C.c_void_p(hout))


@pytest.mark.parametrize("between", ["sync", "totals", "upload"])
def test_download_end_after_a_continuation_elsewhere(oracle, between):
    """A host-side continuation that another entry point performs between hc_download_begin and hc_download_end (every one
    that finishes the runs in flight does: hc_sync, hc_hysteresis_totals -- which may also reset the counter --, hc_upload)
    rewrites maps whose copy is already queued: hc_download_end must copy them again (round-3 advisor finding)."""
    lib = api.load_library()
    w, h = 1000, 2300
    line = np.zeros((h, w), np.uint8)
    line[:, 100:140] = 20
    for r in range(20):
        line[r, 100:140] = 120 - 5 * r
    frames = np.stack([line, synth.natural(w, h, 79)])
    want = oracle.canny_r_batch(frames, 10, 40, threads=4)
    hout = lib.hc_host_alloc(2 * w * h)
    try:
        with api.Context(w, h, 1, 2) as ctx:
            ctx.set_tuning(0, 1)   # one launch queued, a frame that needs 18: continued from the host
            ctx.hysteresis_totals(reset=True)
            ctx.upload(frames)
            ctx.run(api.CannyStage.HYSTER, 2)
            api._ck(lib.hc_download_begin(ctx.handle, C.c_void_p(hout), w, w * h, 2))
            if between == "sync":
                ctx.sync()
            elif between == "totals":
                assert ctx.hysteresis_totals(reset=True)[1] == 1, "the one-launch run was not continued from the host"
            else:
                ctx.upload(frames)
            api._ck(lib.hc_download_end(ctx.handle))
            got = np.ctypeslib.as_array((C.c_uint8 * (2 * w * h)).from_address(hout)).reshape(2, h, w)
            for f in range(2):
                _diff(got[f], want[f], f"download_begin, hc_{between}, download_end: frame {f}")
    finally:
        lib.hc_host_free(C.c_void_p(hout))


def test_product_library_refuses_the_round1_front_forms():
    """HC_OPT_FRONT_SPLIT 1 / 0 in mode R name kernels the product library no longer contains: refused, not ignored (mode O
    keeps its 4-px kernel)."""
    with api.Context(64, 64, 1, 1) as ctx:
        for v in (0, 1):
            with pytest.raises(api.HipCannyError):
                ctx.set_option(api.OPT_FRONT_SPLIT, v)
        ctx.set_option(api.OPT_FRONT_SPLIT, 2)
    with api.Context(64, 64, 1, 1, mode=api.MODE_O) as ctx:
        ctx.set_option(api.OPT_FRONT_SPLIT, 0)
