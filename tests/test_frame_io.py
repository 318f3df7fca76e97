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
    """PGM / PPM parsing (comments, RGB -> BGR), pitched writes and the rejected variants: no GPU involved."""
    _make()
    out = subprocess.run([UNIT, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


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
