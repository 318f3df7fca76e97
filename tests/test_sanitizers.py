"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's template only
offers the switches, all OFF, cmake/Sanitizers.cmake:13-64).  CPU only -- device code cannot be sanitized on this pool.

* the oracle: every entry point the parity tests use, on 1x1 ... 67x45 frames in exact-size heap blocks; the sanitized
  build must print the same checksums as the regular one;
* the image readers of cvp::io (PNG / PGM / PPM parse files of unknown origin): valid files decode to the expected
  pixels, and several hundred truncated / structurally mutated files (chunk CRCs recomputed, so that the mutations reach
  the parser) are either decoded or rejected -- never a sanitizer report, a crash or an allocation the file does not justify."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=0:max_allocation_size_mb=512", UBSAN_OPTIONS="print_stacktrace=1")


def _cc(cmd):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, " ".join(cmd) + "\n" + out.stdout + out.stderr


def test_oracle_under_asan_ubsan(tmp_path):
    src = [os.path.join(ROOT, "tests", "cpp", "san_oracle.c"), os.path.join(ROOT, "oracle", "canny_oracle.c")]
    flags = ["-std=c99", "-ffp-contract=off", "-fno-fast-math", "-fopenmp"]
    _cc(["gcc", *flags, *SAN, "-o", str(tmp_path / "san"), *src, "-lm"])
    _cc(["gcc", *flags, "-O2", "-o", str(tmp_path / "plain"), *src, "-lm"])
    a = subprocess.run([str(tmp_path / "san")], capture_output=True, text=True, timeout=600, env=ENV)
    assert a.returncode == 0 and a.stdout.rstrip().endswith("done"), a.stdout[-2000:] + a.stderr[-4000:]
    b = subprocess.run([str(tmp_path / "plain")], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0
    assert a.stdout == b.stdout and a.stdout.count("\n") > 300


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _png(w, h, ctype, rows, filters=None, plte=None, depth=8, interlace=0, idat_split=1, raw_override=None, end=True):
    """rows: h byte strings of w * samples each (already in file order)"""
    raw = b"".join(bytes([filters[r % len(filters)] if filters else 0]) + rows[r] for r in range(len(rows)))
    if raw_override is not None:
        raw = raw_override
    z = zlib.compress(raw)
    parts = [z[i * len(z) // idat_split:(i + 1) * len(z) // idat_split] for i in range(idat_split)]
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if plte is not None:
        out += _chunk(b"PLTE", plte)
    for p in parts:
        out += _chunk(b"IDAT", p)
    return out + (_chunk(b"IEND", b"") if end else b"")


def test_image_readers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_frameio")
    # frame_io.cpp also holds FrameStreamer, which calls the C ABI: link the (uninstrumented) product library for those symbols
    _cc(["g++", "-std=c++17", *SAN, "-o", exe, os.path.join(ROOT, "tests", "cpp", "san_frameio.cpp"), os.path.join(ROOT, "cudacam_amd", "csrc", "frame_io.cpp"),
         "-L" + os.path.join(ROOT, "cudacam_amd"), "-lhipcanny", "-lz", "-Wl,-rpath," + os.path.join(ROOT, "cudacam_amd")])
    rng = np.random.default_rng(5)
    w, h = 13, 7
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    idx = rng.integers(0, 5, (h, w), dtype=np.uint8)
    pal = rng.integers(0, 256, (5, 3), dtype=np.uint8)
    files, expect = [], []

    def add(name, data, want):
        p = tmp_path / name
        p.write_bytes(data)
        files.append(str(p))
        expect.append(want)

    def fnv(a):
        hsh = 1469598103934665603
        for b in a.tobytes():
            hsh = ((hsh ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % hsh

    bgr = rgb[:, :, ::-1]
    ok_grey, ok_bgr = "ok %d %d 1 %s" % (w, h, fnv(grey)), "ok %d %d 3 %s" % (w, h, fnv(np.ascontiguousarray(bgr)))
    ok_pal = "ok %d %d 3 %s" % (w, h, fnv(np.ascontiguousarray(pal[idx][:, :, ::-1])))
    # well-formed files (the filters are applied as declared only for type 0: use "None" rows so that any filter list decodes to known pixels)
    good_grey = _png(w, h, 0, [grey[r].tobytes() for r in range(h)])
    good_rgb = _png(w, h, 2, [rgb[r].tobytes() for r in range(h)], idat_split=3)
    good_pal = _png(w, h, 3, [idx[r].tobytes() for r in range(h)], plte=pal.tobytes())
    pgm = b"P5\n# comment\n%d %d\n255\n" % (w, h) + grey.tobytes()
    ppm = b"P6 %d %d 255\n" % (w, h) + rgb.tobytes()
    add("g.png", good_grey, ok_grey)
    add("c.png", good_rgb, ok_bgr)
    add("p.png", good_pal, ok_pal)
    add("g.pgm", pgm, ok_grey)
    add("c.ppm", ppm, ok_bgr)
    n_good = len(files)
    # every truncation of the small files
    for name, data, whole in [("g.png", good_grey, ok_grey), ("p.png", good_pal, ok_pal), ("g.pgm", pgm, None), ("c.ppm", ppm, None)]:
        for cut in range(0, len(data) - 1, 1 if len(data) < 200 else 3):
            # (a PNG cut inside its IEND chunk still holds every pixel: decoded like the whole file)
            add("t%d_%s" % (cut, name), data[:cut], ("reject", whole) if whole and cut >= len(data) - 12 else "reject")
    # structural mutations, CRCs valid
    rows_g = [grey[r].tobytes() for r in range(h)]
    muts = {
        "huge.png": _png(65535, 65535, 6, rows_g),                      # 17 GB claimed by a 100-byte file
        "wide.png": _png(w + 1, h, 0, rows_g),                          # raster shorter than the header says
        "tall.png": _png(w, h + 1, 0, rows_g),
        "short.png": _png(w, h - 1, 0, rows_g),                         # raster longer than the header says
        "zero_w.png": _png(0, h, 0, rows_g),
        "depth16.png": _png(w, h, 0, rows_g, depth=16),
        "depth1.png": _png(w, h, 0, rows_g, depth=1),
        "ctype5.png": _png(w, h, 5, rows_g),
        "interlaced.png": _png(w, h, 0, rows_g, interlace=1),
        "filter9.png": _png(w, h, 0, rows_g, filters=[9]),
        "nopal.png": _png(w, h, 3, [idx[r].tobytes() for r in range(h)]),
        "shortpal.png": _png(w, h, 3, [idx[r].tobytes() for r in range(h)], plte=pal.tobytes()[:7]),   # indices beyond the palette
        "emptypal.png": _png(w, h, 3, [idx[r].tobytes() for r in range(h)], plte=b""),
        "noidat.png": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + _chunk(b"IEND", b""),
        "emptyidat.png": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", b"") + _chunk(b"IEND", b""),
        "badzlib.png": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", bytes(range(40))) + _chunk(b"IEND", b""),
        "shortihdr.png": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", b"\0\0\0\x0d\0\0\0\x07") + _chunk(b"IDAT", zlib.compress(b"\0" * 100)) + _chunk(b"IEND", b""),
        "lenlies.png": good_grey[:8] + struct.pack(">I", 0x7FFFFFFF) + good_grey[12:],
        "lenlies2.png": good_grey[:33] + struct.pack(">I", 0xFFFFFFF0) + good_grey[37:],
        "huge.pgm": b"P5 1000000 1000000 255\n" + b"\0" * 64,
        "big.pgm": b"P5 99999999999 7 255\n" + b"\0" * 64,
        "maxval.pgm": b"P5 %d %d 65535\n" % (w, h) + grey.tobytes() * 2,
        "neg.pgm": b"P5 -3 7 255\n" + grey.tobytes(),
        "p2.pgm": b"P2 2 2 255\n1 2 3 4\n",
        "comment_eof.pgm": b"P5 # never ends",
        "empty.bin": b"",
        "sig_only.png": b"\x89PNG\r\n\x1a\n",
    }
    for name, data in muts.items():
        add(name, data, "reject")
    # no IEND / a 4-sample filter mix on RGB: still well-formed pixel data
    add("noend.png", _png(w, h, 0, rows_g, end=False), ok_grey)
    # random byte damage inside the compressed stream and the headers, CRCs recomputed: either outcome, no crash
    z = zlib.compress(b"".join(b"\0" + r for r in rows_g))
    for k in range(200):
        zz = bytearray(z)
        for _ in range(1 + k % 3):
            zz[int(rng.integers(0, len(zz)))] = int(rng.integers(0, 256))
        hdr = bytearray(struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
        if k % 4 == 0:
            hdr[int(rng.integers(0, 13))] = int(rng.integers(0, 256))
        add("r%d.png" % k, b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", bytes(hdr)) + _chunk(b"IDAT", bytes(zz)) + _chunk(b"IEND", b""), None)
    out = subprocess.run([exe, *files], capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-6000:]
    lines = out.stdout.splitlines()
    assert len(lines) == len(files)
    for path, want, got in zip(files, expect, lines):
        if isinstance(want, tuple):
            assert got in want, (os.path.basename(path), want, got)
        elif want is not None:
            assert got == want, (os.path.basename(path), want, got)
    assert sum(1 for g in lines[:n_good] if g.startswith("ok")) == n_good
    assert len(files) > 400
