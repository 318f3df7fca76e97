"""ctypes loader for the CPU oracle (oracle/libcanny_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (cudacam_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcanny_oracle.so")

_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)


class _Outputs(C.Structure):
    _fields_ = [(n, _u8p) for n in ("mono", "blur", "grad_disp", "nms", "thresh", "edges")] + [
        ("sumx", _i16p), ("sumy", _i16p)]


def build(force=False):
    src = os.path.join(_HERE, "canny_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcanny_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_dir_bin.restype = C.c_int
        L.orc_dir_bin_float.restype = C.c_int
        L.orc_dir_bin_kernel.restype = C.c_int
        L.orc_grad_trunc_float.restype = C.c_int
        L.orc_grad_trunc_int.restype = C.c_int
        L.orc_grad_float.restype = C.c_float
        L.orc_hysteresis.restype = C.c_long
        L.orc_canny_o.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _u8p]
        L.orc_canny_o_ex.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _u8p, _u8p]
        L.orc_canny_o_batch.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _u8p, C.c_int]
        _lib = L
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _p16(a):
    return a.ctypes.data_as(_i16p)


def _c8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def gauss_coeffs():
    g = np.zeros(25, np.float32)
    lib().orc_gauss_coeffs(g.ctypes.data_as(C.POINTER(C.c_float)))
    return g.reshape(5, 5)


def gray_bgr(bgr):
    bgr = _c8(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_gray_bgr(_p8(bgr), C.c_size_t(w * 3), w, h, _p8(out), C.c_size_t(w))
    return out


def gaussian(img, fused=True, shortcut=False):
    img = _c8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    if shortcut:
        lib().orc_gaussian_shortcut(_p8(img), C.c_size_t(w), w, h, _p8(out), C.c_size_t(w))
    else:
        lib().orc_gaussian(_p8(img), C.c_size_t(w), w, h, _p8(out), C.c_size_t(w), int(bool(fused)))
    return out


def sobel(blur):
    blur = _c8(blur)
    h, w = blur.shape
    sx = np.empty((h, w), np.int16)
    sy = np.empty((h, w), np.int16)
    lib().orc_sobel(_p8(blur), C.c_size_t(w), w, h, _p16(sx), _p16(sy), C.c_size_t(w))
    return sx, sy


def nms(sx, sy, saturate=False):
    sx = np.ascontiguousarray(sx, np.int16)
    sy = np.ascontiguousarray(sy, np.int16)
    h, w = sx.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_nms(_p16(sx), _p16(sy), C.c_size_t(w), w, h, _p8(out), C.c_size_t(w), int(bool(saturate)))
    return out


def threshold(nms_img, low, high):
    nms_img = _c8(nms_img)
    h, w = nms_img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_threshold(_p8(nms_img), C.c_size_t(w), w, h, int(low), int(high), _p8(out), C.c_size_t(w))
    return out


def hysteresis(thr):
    thr = _c8(thr)
    h, w = thr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_hysteresis(_p8(thr), C.c_size_t(w), w, h, _p8(out), C.c_size_t(w))
    return out


def hysteresis_tiled(thr, tile=30, max_extra=100):
    thr = _c8(thr)
    h, w = thr.shape
    out = np.empty((h, w), np.uint8)
    n = C.c_int(0)
    lib().orc_hysteresis_tiled(_p8(thr), C.c_size_t(w), w, h, _p8(out), C.c_size_t(w), tile, max_extra, C.byref(n))
    return out, n.value


def canny_r(img, low=10, high=40, stages=False, saturate=False):
    """Mode R pipeline.  img: (h,w) u8 mono or (h,w,3) u8 BGR.  Returns edges, or a dict of every
    stage output when stages=True."""
    img = _c8(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    res = {k: np.empty((h, w), np.uint8) for k in ("mono", "blur", "grad_disp", "nms", "thresh", "edges")}
    res["sumx"] = np.empty((h, w), np.int16)
    res["sumy"] = np.empty((h, w), np.int16)
    o = _Outputs()
    for k in ("mono", "blur", "grad_disp", "nms", "thresh", "edges"):
        setattr(o, k, _p8(res[k]))
    o.sumx = _p16(res["sumx"])
    o.sumy = _p16(res["sumy"])
    rc = lib().orc_canny_r(_p8(img), C.c_size_t(w * ch), w, h, ch, int(low), int(high), int(bool(saturate)), C.byref(o))
    if rc:
        raise ValueError("orc_canny_r failed")
    return res if stages else res["edges"]


def canny_r_batch(frames, low=10, high=40, threads=1):
    frames = _c8(frames)
    n, h, w = frames.shape
    out = np.empty((n, h, w), np.uint8)
    rc = lib().orc_canny_r_batch(_p8(frames), w, h, n, int(low), int(high), _p8(out), int(threads))
    if rc:
        raise ValueError("orc_canny_r_batch failed")
    return out


def canny_o(img, low=50, high=150, l2gradient=False):
    img = _c8(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.empty((h, w), np.uint8)
    rc = lib().orc_canny_o(_p8(img), w * ch, w, h, ch, float(low), float(high), int(bool(l2gradient)), _p8(out))
    if rc:
        raise ValueError("orc_canny_o failed")
    return out


def canny_o_stages(img, low=50, high=150, l2gradient=False):
    """(edges, premap): premap is the map before the flood -- 255 seed, 128 candidate, 0 none."""
    img = _c8(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.empty((h, w), np.uint8)
    pre = np.empty((h, w), np.uint8)
    rc = lib().orc_canny_o_ex(_p8(img), w * ch, w, h, ch, float(low), float(high), int(bool(l2gradient)), _p8(out), _p8(pre))
    if rc:
        raise ValueError("orc_canny_o_ex failed")
    return out, pre


def canny_o_batch(frames, low=50, high=150, l2gradient=False, threads=1):
    frames = _c8(frames)
    n, h, w = frames.shape
    out = np.empty((n, h, w), np.uint8)
    rc = lib().orc_canny_o_batch(_p8(frames), w, h, n, float(low), float(high), int(bool(l2gradient)), _p8(out), int(threads))
    if rc:
        raise ValueError("orc_canny_o_batch failed")
    return out
