"""Pins the oracle to the REFERENCE ITSELF: the reference's own device kernels
(src/cvp/cannyEdgeD.cu, compiled in place by oracle/build_ref.sh into oracle/_ref/) are run on the
GPU and compared, stage by stage, with the CPU restatement.  Nothing here reads /root/reference at
run time (the GPU box does not have it); the prebuilt .so files travel with the repo."""
import ctypes as C
import os

import numpy as np
import pytest

from cudacam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p, f32p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int)


def _load(variant):
    path = os.path.join(ROOT, "oracle", "_ref", f"libref_{variant}.so")
    if not os.path.exists(path):
        # the committed fixtures (tests/golden/ref_kernels_*.npz, checked by tests/test_oracle_golden_ref.py on CPU) pin
        # the same kernels' outputs; with neither the binaries nor the fixtures the oracle would be unpinned: fail
        gold = os.path.join(ROOT, "tests", "golden", f"ref_kernels_{variant}.npz")
        assert os.path.exists(gold), f"neither {path} nor {gold} exists: run oracle/build_ref.sh (needs /root/reference)"
        pytest.skip(f"{path} not built here; its outputs are pinned by {os.path.basename(gold)}")
    from cudacam_amd import api
    api.preload_hip_runtime()  # one HIP runtime per process (see api.preload_hip_runtime)
    L = C.CDLL(path)
    L.ref_run.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, f32p, f32p, f32p, f32p, u8p, u8p, u8p, u8p, ip]
    L.ref_hysteresis.argtypes = [u8p, C.c_int, C.c_int, u8p, ip]
    return L


def ref_run(L, img, low=10, high=40):
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else 3
    h, w = img.shape[:2]
    o = {k: np.zeros((h, w), np.uint8) for k in ("mono", "blur", "grad_disp", "nms", "thresh", "edges")}
    fl = {k: np.zeros((h, w), np.float32) for k in ("sobelX", "sobelY", "grad", "slope")}
    n = C.c_int(0)
    p8 = lambda a: a.ctypes.data_as(u8p)
    pf = lambda a: a.ctypes.data_as(f32p)
    rc = L.ref_run(p8(img), w, h, ch, low, high, p8(o["mono"]), p8(o["blur"]), pf(fl["sobelX"]), pf(fl["sobelY"]), pf(fl["grad"]),
                   pf(fl["slope"]), p8(o["grad_disp"]), p8(o["nms"]), p8(o["thresh"]), p8(o["edges"]), C.byref(n))
    assert rc == 0
    o.update(fl)
    o["launches"] = n.value
    return o


# pairs (sumX, sumY) where the float atan2 binning may legitimately differ from the integer rule
# (angles within a few ulp of a 22.5-degree boundary): sqrt(2) convergents
def _near_boundary(sx, sy):
    a, b = abs(int(sx)), abs(int(sy))
    lo, hi = min(a, b), max(a, b)
    return hi > 0 and abs(lo * 985 - hi * 408) <= 8 * hi // 985 + 8


IMAGES = [
    ("natural_640x480", lambda: synth.natural(640, 480, 1)),
    ("noise_333x222", lambda: synth.noise(333, 222, 2)),
    ("flat100", lambda: synth.flat(40, 36, 100)),
    ("step255", lambda: synth.steps(96, 64, 255, "vertical")),
    ("step234d", lambda: synth.steps(96, 96, 234, "diagonal")),
    ("serpentine", lambda: synth.serpentine(300, 200)),
]


@pytest.mark.parametrize("variant,fused", [("fma", True), ("nofma", False)])
@pytest.mark.parametrize("name,make", IMAGES, ids=[n for n, _ in IMAGES])
def test_reference_kernels_vs_oracle(oracle, variant, fused, name, make):
    L = _load(variant)
    img = make()
    r = ref_run(L, img)
    blur = oracle.gaussian(img, fused=fused)
    assert np.array_equal(r["blur"], blur), f"blur differs at {np.argwhere(r['blur'] != blur)[:5]}"
    sx, sy = oracle.sobel(blur)
    assert np.array_equal(r["sobelX"] * 8, sx.astype(np.float32)) and np.array_equal(r["sobelY"] * 8, sy.astype(np.float32))
    # hipcc turns the reference's `min((unsigned char)gradVal, 255)` into a saturating store (see
    # canny_oracle.c orc_nms): compare with the oracle's saturate variant; everything else is shared
    nms = oracle.nms(sx, sy, saturate=True)
    bad = np.argwhere(r["nms"] != nms)
    for (y, x) in bad:  # only direction-boundary pairs may differ (libm atan2 vs exact rule)
        assert _near_boundary(sx[y, x], sy[y, x]), (y, x, int(sx[y, x]), int(sy[y, x]), int(r["nms"][y, x]), int(nms[y, x]))
    # later stages from the reference's own NMS plane, so that a direction-boundary pixel cannot hide a difference
    thr = oracle.threshold(r["nms"], 10, 40)
    assert np.array_equal(r["thresh"], thr)
    assert np.array_equal(r["edges"], oracle.hysteresis(thr))
    assert np.array_equal(r["grad_disp"], oracle.canny_r(img, stages=True)["grad_disp"]) or not fused
    if fused:  # and the product itself, saturate option on, equals the reference kernels bit for bit
        from cudacam_amd import api
        h, w = img.shape
        ok = np.ones((h, w), bool)
        for (y, x) in bad:  # a boundary pixel's own NMS value, and whatever the hysteresis grows from it, is excused
            ok[max(0, y - 1):y + 2, max(0, x - 1):x + 2] = False
        with api.Context(w, h, 1, 1) as ctx:
            ctx.set_option(api.OPT_NMS_SATURATE, 1)
            ctx.set_option(api.OPT_DEBUG_TAPS, 1)
            for st, key in ((api.CannyStage.GAUSSIAN, "blur"), (api.CannyStage.NMS, "nms"), (api.CannyStage.THRESH, "thresh"),
                            (api.CannyStage.HYSTER, "edges")):
                got = ctx.process(img, st)[0]
                if key in ("blur",) or len(bad) == 0:
                    assert np.array_equal(got, r[key]), f"product vs reference kernels: {key}"
                elif key != "edges":
                    assert np.array_equal(got[ok], r[key][ok]), f"product vs reference kernels: {key} (boundary pixels masked)"
            # the fast path's own intermediates against the reference kernels' planes
            assert np.array_equal(ctx.debug_tap(api.TAP_BLUR)[0], r["blur"]), "fast-path blur vs reference gaussianFilter5x5"
            tt = ctx.debug_tap(api.TAP_THRESH)[0]
            assert np.array_equal(tt[ok], r["thresh"][ok]), "fast-path bit planes vs reference doubleThreshold"
    assert r["launches"] >= 1


def test_reference_bgr_gray(oracle):
    L = _load("fma")
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (50, 77, 3), dtype=np.uint8)
    r = ref_run(L, img)
    assert np.array_equal(r["mono"], oracle.gray_bgr(img))
    assert np.array_equal(r["edges"], oracle.canny_r(img, saturate=True))


def test_reference_hysteresis_cap(oracle):
    """The reference stops after 101 launches (cannyEdgeH.cu:312-324); the oracle's tiled emulation
    reproduces that truncation, the full fixpoint differs only on such adversarial maps."""
    L = _load("fma")
    t = synth.thresh_map_random(97, 61, 1, 0.45, 0.01)
    out = np.zeros_like(t)
    n = C.c_int()
    assert L.ref_hysteresis(t.ctypes.data_as(u8p), 97, 61, out.ctypes.data_as(u8p), C.byref(n)) == 0
    assert np.array_equal(out, oracle.hysteresis(t))
    s = synth.thresh_map_serpentine(200, 300)
    out = np.zeros_like(s)
    assert L.ref_hysteresis(s.ctypes.data_as(u8p), 200, 300, out.ctypes.data_as(u8p), C.byref(n)) == 0
    assert n.value == 101
    full = oracle.hysteresis(s)
    assert (out != full).any() and ((out == 255) <= (full == 255)).all()  # truncated subset of the fixpoint
