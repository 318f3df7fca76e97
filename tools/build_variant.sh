#!/usr/bin/env bash
# Kernel experiments: builds cudacam_amd/libhipcanny_<name>.so with extra compiler flags (e.g. -DF8_X_NOFIX); select it
# with HIPCANNY_LIB=<path> (cudacam_amd/api.py).  Timing experiments only -- such builds may compute wrong results.
set -euo pipefail
name="$1"; shift
cd "$(dirname "$0")/../cudacam_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function "$@" \
  csrc/canny_kernels.hip csrc/front8.hip csrc/hipcanny.hip -o "libhipcanny_${name}.so"
echo "cudacam_amd/libhipcanny_${name}.so"
