"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/hipcanny.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from cudacam_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return api.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "hipcanny.h")).read()
    declared = sorted(set(re.findall(r"\b(hc_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(api.ABI_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.hc_version()
    assert isinstance(api.last_error(), str)


def test_bad_arguments_rejected(lib):
    assert not lib.hc_create(0, 0, 10, 1, 1, 0)
    assert "bad" in api.last_error()
    assert not lib.hc_create(0, 10, 10, 2, 1, 0)       # only 1 or 3 channels (CV_8UC1 / CV_8UC3)
    assert lib.hc_set_thresholds(None, 1, 2) != 0


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.HipCannyError):
        api.Context(64, 64)
    with pytest.raises(api.HipCannyError):
        api.cvPipeline(0, 64, 64, 1)


def test_product_does_not_touch_oracle():
    """The product path must never import / link the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cudacam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "canny_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "canny_oracle" not in open(os.path.join(dirpath, f), errors="ignore").read()
    # tools/ (benchmark helpers, profiling scripts, the file front end) stay clear of it too
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith((".py", ".sh", ".cpp", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "canny_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_stage_names():
    assert api.CANNY_STAGES[api.CannyStage.MONO] == "1/6 Mono Conversion"
    assert api.CANNY_STAGES[api.CannyStage.HYSTER] == "6/6 Hysteresis"
    assert [int(s) for s in api.CannyStage] == [0, 1, 2, 3, 4, 5]


def test_timer_manager():
    tm = api.TimerManager.Get()
    tm.createTimer("t")
    tm.addTime("t", 2.0)
    tm.addTime("t", 4.0)
    assert tm.getAverageTime("t") == 3.0
    assert tm.getAverageTime("missing") == 0.0


def test_round1_front_kernels_are_not_in_the_product_library(lib):
    """The round-1 front kernels of Mode R (k_front, k_blur, k_nms: legacy_front.hip) are test infrastructure since round 3:
    the product library does not contain them; libhipcanny_legacy.so (the same sources + those kernels) does."""
    product = open(api.LIB_PATH, "rb").read()
    legacy = open(build.build_legacy(), "rb").read()
    for name in (b"_ZN2hc6k_blurILi", b"_ZN2hc5k_nmsE", b"_ZN2hc7k_frontILi"):   # the kernels' mangled names
        assert name not in product, name
        assert name in legacy, name
    for name in (b"k_front8", b"k_front8o", b"k_front_o", b"k_hyst"):
        assert name in product, name
    assert b"test build" not in lib.hc_version()
    assert b"test build" in api.load_library(legacy=True).hc_version()
