#!/usr/bin/env bash
# Kernel experiments: builds cudacam_amd/exp/libhipcanny_<name>.so with extra compiler flags for ONE source file (e.g.
# tools/build_variant.sh abl1 front_mx.hip -DMX_ABL=1); select it with HIPCANNY_LIB=<path> (cudacam_amd/api.py).  The
# other sources are compiled once into cudacam_amd/exp/obj/.  Timing experiments only -- such builds may compute wrong results.
set -euo pipefail
name="$1"; src="$2"; shift 2
cd "$(dirname "$0")/../cudacam_amd"
mkdir -p exp/obj
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
objs=()
for f in canny_kernels.hip front8.hip front_mx.hip hipcanny.hip; do
  if [ "$f" = "$src" ]; then
    $CC "$@" -c "csrc/$f" -o "exp/obj/${f%.hip}_${name}.o"
    objs+=("exp/obj/${f%.hip}_${name}.o")
  else
    o="exp/obj/${f%.hip}.o"
    if [ ! -f "$o" ] || [ "csrc/$f" -nt "$o" ] || [ csrc/canny_common.h -nt "$o" ] || [ csrc/canny_device.h -nt "$o" ] || [ ../include/hipcanny.h -nt "$o" ]; then $CC -c "csrc/$f" -o "$o"; fi
    objs+=("$o")
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "exp/libhipcanny_${name}.so"
echo "cudacam_amd/exp/libhipcanny_${name}.so"
