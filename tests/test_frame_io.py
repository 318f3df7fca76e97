"""Frame I/O either side of the hot path: cvp::io PNM files and the FrameStreamer ring,
through the headless front end tools/bin/canny_files."""
import json
import os
import subprocess

import numpy as np
import pytest

from cudacam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "bin", "canny_files")
UNIT = os.path.join(ROOT, "tests", "cpp", "test_frameio")


def _make():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cudacam_amd", "csrc")])


def _write_pnm(path, img):
    with open(path, "wb") as f:
        if img.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            f.write(img.tobytes())
        else:   # memory order B, G, R -> RGB on disk
            f.write(b"P6\n# written by the test\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            f.write(np.ascontiguousarray(img[:, :, ::-1]).tobytes())


def _read_pgm(path):
    raw = open(path, "rb").read()
    assert raw[:2] == b"P5"
    head = raw.split(b"\n", 3)
    w, h = map(int, head[1].split())
    assert head[2] == b"255"
    return np.frombuffer(head[3], np.uint8, w * h).reshape(h, w)


def test_pnm_reader_writer(tmp_path):
    """PGM / PPM parsing (comments, RGB -> BGR), PNG round trips, pitched writes and the rejected variants: no GPU involved."""
    _make()
    out = subprocess.run([UNIT, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def _png_bytes(img, filters):
    """An independent PNG encoder (numpy + zlib) that exercises every scanline filter type: `filters[r % len]` on row r."""
    import struct
    import zlib
    h, w = img.shape[:2]
    spp = 1 if img.ndim == 2 else img.shape[2]
    rows = img.reshape(h, w * spp).astype(np.int32) if spp != 3 else np.ascontiguousarray(img[:, :, ::-1]).reshape(h, w * 3).astype(np.int32)   # B,G,R -> RGB
    raw = bytearray()
    prev = np.zeros(w * spp, np.int32)
    for r in range(h):
        x = rows[r]
        a = np.concatenate([np.zeros(spp, np.int32), x[:-spp]])
        c = np.concatenate([np.zeros(spp, np.int32), prev[:-spp]])
        ft = filters[r % len(filters)]
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (a + prev) // 2
        else:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw.append(ft)
        raw += ((x - pred) & 255).astype(np.uint8).tobytes()
        prev = x

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6, 2: 4}[spp], 0, 0, 0)
    z = zlib.compress(bytes(raw), 9)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")


@pytest.mark.gpu
def test_png_files_all_filters_match_oracle(oracle, tmp_path):
    """BASELINE configs[0]'s input format: PNG files (every scanline filter, split IDAT, ancillary chunk, RGBA) through
    the headless front end; the maps equal the oracle's on the decoded pixels."""
    _make()
    w, h = 324, 200
    grey = synth.natural(w, h, 321)
    bgr = np.stack([synth.natural(w, h, 400 + c) for c in range(3)], axis=-1)
    rgba_mem = np.concatenate([bgr[:, :, ::-1], np.full((h, w, 1), 200, np.uint8)], axis=-1)   # RGBA as stored in the file
    (tmp_path / "g.png").write_bytes(_png_bytes(grey, [0, 1, 2, 3, 4]))
    (tmp_path / "g4.png").write_bytes(_png_bytes(grey, [4]))
    out = subprocess.run([CLI, "-o", str(tmp_path), "--batch", "2", str(tmp_path / "g.png"), str(tmp_path / "g4.png")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    want = oracle.canny_r(grey, 10, 40)
    assert np.array_equal(_read_pgm(str(tmp_path / "g.edges.pgm")), want) and np.array_equal(_read_pgm(str(tmp_path / "g4.edges.pgm")), want)
    # 3-channel: RGB and RGBA files decode to the same B,G,R frame
    import struct, zlib  # noqa: E401,F401
    (tmp_path / "c.png").write_bytes(_png_bytes(bgr, [1, 4, 3]))
    rows = rgba_mem.reshape(h, w * 4)
    raw = b"".join(b"\0" + rows[r].tobytes() for r in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    (tmp_path / "a.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    out = subprocess.run([CLI, "-o", str(tmp_path), "--batch", "2", str(tmp_path / "c.png"), str(tmp_path / "a.png")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    want3 = oracle.canny_r(bgr, 10, 40)
    assert np.array_equal(_read_pgm(str(tmp_path / "c.edges.pgm")), want3) and np.array_equal(_read_pgm(str(tmp_path / "a.edges.pgm")), want3)


@pytest.mark.gpu
@pytest.mark.parametrize("channels,batch", [(1, 2), (1, 16), (3, 3)])
def test_streamed_files_match_oracle(oracle, tmp_path, channels, batch):
    """7 frames on disk -> FrameStreamer (3 slots, page-locked staging, overlapped transfers) -> 7 edge maps,
    in order, bit-exact with the oracle; a ragged last batch included."""
    _make()
    w, h, n = 324, 200, 7
    frames, paths = [], []
    for i in range(n):
        img = synth.natural(w, h, 100 + i) if channels == 1 else np.stack([synth.natural(w, h, 100 + i + 50 * c) for c in range(3)], axis=-1)
        frames.append(img)
        paths.append(str(tmp_path / (f"f{i:02d}." + ("pgm" if channels == 1 else "ppm"))))
        _write_pnm(paths[-1], img)
    out = subprocess.run([CLI, "-o", str(tmp_path), "--low", "10", "--high", "40", "--batch", str(batch), "--repeat", "2"] + paths,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["frames"] == 2 * n and rep["written"] == n
    for i in range(n):
        got = _read_pgm(str(tmp_path / f"f{i:02d}.edges.pgm"))
        assert np.array_equal(got, oracle.canny_r(frames[i], 10, 40)), f"frame {i}"


@pytest.mark.gpu
def test_baseline_config0_png_640x480_mode_o(oracle, tmp_path):
    """BASELINE configs[0]: a single 640x480 grayscale PNG through cv::Canny -- here the PNG goes through the headless
    front end in Mode O (cv::Canny semantics) and Mode R; the maps equal the CPU restatement / the oracle.  (A real OpenCV
    is probed by bench.py's cpu_baseline leg; this image has none.)"""
    _make()
    img = synth.natural(640, 480, 640480)
    (tmp_path / "frame.png").write_bytes(_png_bytes(img, [4, 2, 1, 0, 3]))
    out = subprocess.run([CLI, "-o", str(tmp_path), "--mode", "O", "--batch", "1", str(tmp_path / "frame.png")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert np.array_equal(_read_pgm(str(tmp_path / "frame.edges.pgm")), oracle.canny_o(img, 50, 150))
    out = subprocess.run([CLI, "-o", str(tmp_path), "--batch", "1", str(tmp_path / "frame.png")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert np.array_equal(_read_pgm(str(tmp_path / "frame.edges.pgm")), oracle.canny_r(img, 10, 40))
