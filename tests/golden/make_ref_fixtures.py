"""Generates tests/golden/ref_kernels_{fma,nofma}.npz: inputs and EVERY stage output of the reference's own device
kernels (src/cvp/cannyEdgeD.cu compiled in place by oracle/build_ref.sh into oracle/_ref/libref_*.so), run on an MI355X.

Run once on the GPU box (the reference sources never travel; the prebuilt oracle/_ref/*.so do):
    gpurun -- python tests/golden/make_ref_fixtures.py gpurun_out/fixtures
and copy the two .npz files into tests/golden/.  The files hold data only (u8 / i16 / f32 arrays).

What the fixtures pin and what they cannot: the kernels are the reference's source text, but compiled by hipcc for
gfx950 (no nvcc in this image) and launched by oracle/ref_driver.hip, a restatement of the host launch sequence of
src/cvp/cannyEdgeH.cu.  hipcc lowers `min((unsigned char)gradVal, 255)` (cannyEdgeD.cu:267) to a saturating store, so
the `nms` arrays here follow the oracle's saturate=True variant; everything else is common to both readings.
`fma` = default contraction forced to fmaf (nvcc's -fmad=true), `nofma` = -ffp-contract=off."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cudacam_amd import api, synth  # noqa: E402

u8p, f32p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int)


def load(variant):
    api.preload_hip_runtime()
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_{variant}.so"))
    L.ref_run.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, f32p, f32p, f32p, f32p, u8p, u8p, u8p, u8p, ip]
    L.ref_hysteresis.argtypes = [u8p, C.c_int, C.c_int, u8p, ip]
    return L


def ref_run(L, img, low, high):
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
    o["launches"] = np.int32(n.value)
    return o


def cases():
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (50, 77, 3), dtype=np.uint8)
    bgr2 = np.stack([synth.natural(120, 90, 9), synth.natural(120, 90, 10), synth.natural(120, 90, 11)], axis=-1)
    return [
        ("natural_160x120", synth.natural(160, 120, 1), 10, 40),
        ("natural_160x120_t60_200", synth.natural(160, 120, 1), 60, 200),
        ("noise_97x61", synth.noise(97, 61, 2), 10, 40),
        ("flat100_40x36", synth.flat(40, 36, 100), 10, 40),
        ("flat255_33x31", synth.flat(33, 31, 255), 10, 40),
        ("step255_v_96x64", synth.steps(96, 64, 255, "vertical"), 10, 40),
        ("step240_h_64x64", synth.steps(64, 64, 240, "horizontal"), 10, 40),
        ("step234_d_96x96", synth.steps(96, 96, 234, "diagonal"), 10, 40),
        ("step100_v_64x48", synth.steps(64, 48, 100, "vertical"), 10, 40),
        ("serpentine_150x100", synth.serpentine(150, 100), 10, 40),
        ("one_px", np.array([[200]], np.uint8), 10, 40),
        ("noise_5x5", synth.noise(5, 5, 4), 10, 40),
        ("bgr_noise_77x50", bgr, 10, 40),
        ("bgr_natural_120x90", bgr2, 10, 40),
    ]


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    for variant in ("fma", "nofma"):
        L = load(variant)
        d = {}
        names = []
        for name, img, low, high in cases():
            r = ref_run(L, img, low, high)
            names.append(name)
            d[name + "/input"] = img
            d[name + "/thresholds"] = np.array([low, high], np.int32)
            for k in ("mono", "blur", "grad_disp", "nms", "thresh", "edges", "grad", "slope", "launches"):
                d[name + "/" + k] = r[k]
            # sobelXY stores sum / 8.0f (cannyEdgeD.cu:168-169): exact in binary32, kept as the integer sums
            for k in ("sobelX", "sobelY"):
                v = r[k] * 8
                assert np.array_equal(v, np.round(v)) and np.abs(v).max() <= 1020
                d[name + "/" + k + "8"] = v.astype(np.int16)
        # hysteresis + removeCandidates alone (cannyEdgeD.cu:295-395, launch loop cannyEdgeH.cu:297-338)
        for hname, t in (("hyst_random_97x61", synth.thresh_map_random(97, 61, 1, 0.45, 0.01)),
                         ("hyst_random_200x90", synth.thresh_map_random(200, 90, 2, 0.30, 0.004)),
                         ("hyst_serpentine_200x300", synth.thresh_map_serpentine(200, 300))):
            out = np.zeros_like(t)
            n = C.c_int()
            assert L.ref_hysteresis(t.ctypes.data_as(u8p), t.shape[1], t.shape[0], out.ctypes.data_as(u8p), C.byref(n)) == 0
            names.append(hname)
            d[hname + "/input"] = t
            d[hname + "/edges"] = out
            d[hname + "/launches"] = np.int32(n.value)
        d["names"] = np.array(names)
        path = os.path.join(outdir, f"ref_kernels_{variant}.npz")
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fixtures"))
