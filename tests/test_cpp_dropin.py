"""The C++ drop-in surface (include/cvp/*.hpp over the C ABI)."""
import os
import subprocess

import numpy as np
import pytest

from cudacam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cudacam_amd", "libcvProcessing_hip.so")
BIN = os.path.join(ROOT, "tests", "cpp", "test_cvpipeline")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cudacam_amd", "csrc")])


def test_cpp_surface_builds_and_exports():
    """CPU: the header-compatible classes compile against include/cvp and export the reference's members."""
    from cudacam_amd import build
    build.build()
    _build()
    syms = subprocess.check_output(["nm", "-DC", "--defined-only", LIB], text=True)
    # cv::Mat never crosses the boundary (include/cvp/frameView.hpp): the Mat-taking members are inline
    # wrappers in the headers, the library exports the view forms and no symbol that mentions cv::Mat
    assert "cv::Mat" not in syms, [l for l in syms.splitlines() if "cv::Mat" in l]
    for want in ("cvp::cvPipeline::processView(cvp::FrameView const&, cvp::CannyStage)", "cvp::cvPipeline::setLowThreshold(unsigned char)",
                 "cvp::cvPipeline::getHighThreshold() const", "cvp::cvPipeline::enableCudaProfiling(bool)",
                 "cvp::cvPipeline::isCudaProfilingEnabled() const", "cvp::cuda::CannyEdge::runView(cvp::FrameView const&, cvp::CannyStage)",
                 "cvp::cuda::CannyEdge::CannyEdge(unsigned int, unsigned int, unsigned int, int)"):
        assert want in syms, want


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["test_cvpipeline", "test_cvpipeline_othermat"])
@pytest.mark.parametrize("channels,stage", [(1, 5), (3, 5), (1, 3), (3, 2), (3, 0), (1, 0), (3, 4)])
def test_cpp_pipeline_matches_oracle(oracle, tmp_path, channels, stage, binary):
    """`test_cvpipeline_othermat` is the same driver built against a cv::Mat with OpenCV's member order
    (tests/cpp/other_mat_layout): the host's Mat layout must not matter to the prebuilt library."""
    BIN = os.path.join(ROOT, "tests", "cpp", binary)
    if not os.path.exists(BIN):
        _build()
    w, h = 321, 200
    img = synth.natural(w, h, 77) if channels == 1 else np.random.default_rng(1).integers(0, 256, (h, w, 3), dtype=np.uint8)
    fin, fout = tmp_path / "in.raw", tmp_path / "out.raw"
    img.tofile(fin)
    r = subprocess.run([BIN, str(w), str(h), str(channels), str(fin), str(stage), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    got = np.fromfile(fout, np.uint8).reshape(h, w)
    key = {5: "edges", 4: "thresh", 3: "nms", 2: "grad_disp", 0: "mono"}[stage]
    if stage == 5 and channels == 3:  # all six timers hold a sample
        assert r.stdout.count("samples 2") == 6, r.stdout
    assert np.array_equal(got, oracle.canny_r(img, 10, 40, stages=True)[key])
