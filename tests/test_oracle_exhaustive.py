"""Exhaustive domain checks: every integer shortcut used by the oracle / HIP kernels against the
literal float formulas of the reference."""
import ctypes as C
import json
import os

import numpy as np

from cudacam_amd import synth

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_kat.json")))


def test_direction_all_pairs(oracle):
    L = oracle.lib()
    mk, mf = C.c_int(), C.c_int()
    pairs = np.zeros((64, 2), np.int16)
    L.orc_check_dir_all(C.byref(mk), C.byref(mf), pairs.ctypes.data_as(C.POINTER(C.c_int16)), 64)
    assert mk.value == 0                      # kernel form == integer rule on all 4,165,680 pairs
    got = sorted(map(tuple, pairs[: mf.value].tolist()))
    assert got == sorted(map(tuple, KAT["direction_float_vs_integer_exceptions_glibc"]))


def test_gradient_all_pairs(oracle):
    assert oracle.lib().orc_check_grad_all() == 0


def test_div159_magic():
    S = np.arange(0, 159 * 255 + 1, dtype=np.uint64)
    assert np.array_equal((S * np.uint64(52759)) >> np.uint64(23), S // np.uint64(159))
    assert int(S[-1]) * 52759 < 2 ** 31
    # k_blur's undecidable-pixel test: S % 159 == 0  <=>  bits 15..22 of S*52759 are all zero (the byte below the
    # quotient byte after >> 15), and the byte-wise zero detector never misses one (it may over-flag, which is harmless)
    P = S * np.uint64(52759)
    frac_byte = (P >> np.uint64(15)) & np.uint64(0xFF)
    assert np.array_equal(frac_byte == 0, S % np.uint64(159) == 0)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint64)
    x[: 4 * 65536] = (rng.integers(0, 256, (65536, 4), dtype=np.uint64) * np.array([1, 0, 1, 1], dtype=np.uint64) << np.array([0, 8, 16, 24], dtype=np.uint64)).sum(1).repeat(4)
    hz = ((x - np.uint64(0x01010101)) & ~x & np.uint64(0x80808080)) & np.uint64(0xFFFFFFFF)
    for b in range(4):
        is_zero = ((x >> np.uint64(8 * b)) & np.uint64(0xFF)) == 0
        flagged = (hz >> np.uint64(8 * b + 7)) & np.uint64(1)
        assert np.all(flagged[is_zero] == 1)


def test_gaussian_shortcut_random_patches(oracle):
    L = oracle.lib()
    L.orc_check_gauss_random.restype = C.c_long
    nd, nm = C.c_long(), C.c_long()
    bad = L.orc_check_gauss_random(C.c_ulonglong(0xC0FFEE), C.c_long(20_000_000), C.byref(nd), C.byref(nm))
    assert bad == 0 and nd.value > 0 and nm.value > nd.value


def test_gaussian_shortcut_images(oracle):
    for img in (synth.noise(321, 123, 7), synth.natural(333, 222, 3), synth.flat(33, 17, 255), synth.flat(20, 20, 0),
                synth.steps(64, 48, 255, "diagonal")):
        assert np.array_equal(oracle.gaussian(img, fused=True), oracle.gaussian(img, shortcut=True))


def test_threshold_bands_all_S():
    """nms value = isqrt(S>>2) & 0xFF (the u8 wrap).  `value > T` as interval tests on S -- the form
    the HIP kernel evaluates with v_cmp masks -- for every reachable S and every T."""
    S = np.arange(0, 2 * 1020 * 1020 + 1, dtype=np.int64)
    g = np.floor(np.sqrt((S >> 2).astype(np.float64))).astype(np.int64)
    g -= (g * g > (S >> 2))
    g += ((g + 1) * (g + 1) <= (S >> 2))
    val = g & 0xFF
    B0, B1 = 4 * 256 * 256, 4 * 512 * 512
    for T in list(range(0, 256, 5)) + [9, 10, 39, 40, 254, 255]:
        a0, a1, a2 = 4 * (T + 1) ** 2, 4 * (257 + T) ** 2, 4 * (513 + T) ** 2
        band = ((S >= a0) & (S < B0)) | ((S >= a1) & (S < B1)) | (S >= a2)
        assert np.array_equal(band, val > T), T


def test_quotient_by_24bit_multiply_all_sums():
    """k_front8's exact quotient (round 3): floor(S / 159) = (S * 105518) >> 24 for every reachable S = sum K*x <= 40545, the
    product stays below 2^32 (one v_mul_u32_u24 per pixel), and its bits 16..23 -- the "fraction byte" -- are all zero exactly
    when S % 159 == 0, the only sums for which the reference's float chain (cannyEdgeD.cu:102-115) can fall below the integer
    quotient (those pixels get the literal chain)."""
    S = np.arange(0, 159 * 255 + 1, dtype=np.uint64)
    P = S * np.uint64(105518)
    assert int(P.max()) < 2 ** 32
    assert np.array_equal(P >> np.uint64(24), S // np.uint64(159))
    assert np.array_equal(((P >> np.uint64(16)) & np.uint64(255)) == 0, S % np.uint64(159) == 0)


def test_mx_biased_quotient_and_toeplitz_sums():
    """k_front_mx (round 4) feeds the i8 MFMA with x ^ 0x80 (= x - 128 as a signed byte), so the matrix pipe returns
    S' = S - 128 * 159; one v_mad_i32_i24 per pixel forms P = S' * 105518 + C0 with C0 = 20352 * 105518 + 2^31 (mod 2^32):
    byte 3 of P is floor(S / 159) ^ 0x80 -- the blur byte in the ring's biased form -- and byte 2 is zero exactly when
    S % 159 == 0, for every reachable S; both factors fit the signed 24-bit operands of the instruction.  And the banded
    Toeplitz form itself: 5 row-times-matrix products with the A matrices of front_mx.hip (output column o <- window columns
    o .. o + 4) give the 5 x 5 integer sums of a random 20 x 32 window exactly (numpy int64 in place of the MFMA's int32)."""
    S = np.arange(0, 159 * 255 + 1, dtype=np.int64)
    Sp = S - 128 * 159
    assert Sp.min() >= -(1 << 23) and Sp.max() < (1 << 23) and 105518 < (1 << 23)
    C0 = (20352 * 105518 + (1 << 31)) % (1 << 32)
    P = (Sp * 105518 + C0) % (1 << 32)
    assert np.array_equal(P, (S * 105518 + (1 << 31)) % (1 << 32))
    assert np.array_equal((P >> 24) ^ 0x80, S // 159)
    assert np.array_equal(((P >> 16) & 255) == 0, S % 159 == 0)
    # the Toeplitz matrices (7 x 32 x 32: blur rows 0/4, 1/3, 2 of the kernel; the layout permutation of the rows is left out)
    K = np.array([[2, 4, 5, 4, 2], [4, 9, 12, 9, 4], [5, 12, 15, 12, 5], [4, 9, 12, 9, 4], [2, 4, 5, 4, 2]], np.int64)
    A = np.zeros((5, 28, 32), np.int64)
    for i in range(5):
        for o in range(28):
            A[i, o, o:o + 5] = K[i]
    rng = np.random.default_rng(5)
    win = rng.integers(0, 256, (20, 32)).astype(np.int64)
    xs = (win ^ 0x80).astype(np.int8).astype(np.int64)   # the bytes as the MFMA reads them
    for r in range(16):
        acc = sum(A[i] @ xs[r + i] for i in range(5))     # 28 sums S' of blur row r + 2
        want = np.array([(K * win[r:r + 5, o:o + 5]).sum() for o in range(28)])
        assert np.array_equal(acc + 128 * 159, want)
    # the Sobel masks sum to zero: the bias drops out (cannyEdgeD.cu:158-167)
    blur = rng.integers(0, 256, (3, 32)).astype(np.int64)
    bs = (blur ^ 0x80).astype(np.int8).astype(np.int64)
    for o in range(1, 31):
        sx = lambda b: -b[0, o - 1] + b[0, o + 1] - 2 * b[1, o - 1] + 2 * b[1, o + 1] - b[2, o - 1] + b[2, o + 1]
        sy = lambda b: b[0, o - 1] + 2 * b[0, o] + b[0, o + 1] - b[2, o - 1] - 2 * b[2, o] - b[2, o + 1]
        assert sx(bs) == sx(blur) and sy(bs) == sy(blur)
