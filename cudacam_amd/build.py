"""Builds libhipcanny.so (the product) for gfx950 with hipcc, in-tree.

Usage: python -m cudacam_amd.build [--force]
hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhipcanny.so")
# the same sources + the round-1 front kernels of Mode R (HC_OPT_FRONT_SPLIT 1 / 0): independent implementations for the
# parity tests, not part of the product
LIB_LEGACY = os.path.join(HERE, "libhipcanny_legacy.so")
SOURCES = ["canny_kernels.hip", "front8.hip", "front_mx.hip", "hipcanny.hip"]
LEGACY_SOURCES = SOURCES + ["legacy_front.hip"]
DEPS = LEGACY_SOURCES + ["canny_common.h", "canny_device.h", os.path.join("..", "..", "include", "hipcanny.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_legacy(force=False, verbose=False):
    """libhipcanny_legacy.so: the product's sources + legacy_front.hip (-DHC_LEGACY_FRONT); test infrastructure."""
    if not force and not stale(LIB_LEGACY):
        return LIB_LEGACY
    cmd = [HIPCC] + FLAGS + ["-DHC_LEGACY_FRONT"] + [os.path.join(CSRC, s) for s in LEGACY_SOURCES] + ["-o", LIB_LEGACY]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_LEGACY


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_legacy(force="--force" in sys.argv, verbose=True))
