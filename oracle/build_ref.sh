#!/usr/bin/env bash
# Builds oracle/_ref/: the REFERENCE's device kernels (src/cvp/cannyEdgeD.cu) compiled IN PLACE from
# /root/reference with hipcc for gfx950, plus the launcher in ref_driver.hip.  Test infrastructure
# only; outputs go to oracle/_ref/ (git-ignored, travels to the GPU box).  No reference source is
# copied.  Needs /root/reference (absent on the GPU box: the prebuilt .so files are used there).
#   libref_fma.so    -fno-slp-vectorize : hipcc then contracts all 25 Gaussian taps into v_fma/v_fmac
#                                         in source order == nvcc's default -fmad=true (checked below)
#   libref_nofma.so  -ffp-contract=off  : separately rounded multiply+add (the x86-emulation variant
#                                         of SURVEY App. C.4)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REFERENCE_ROOT:-/root/reference}"
SRC="$REF/src/cvp"
if [ ! -f "$SRC/cannyEdgeD.cu" ]; then echo "build_ref: $SRC/cannyEdgeD.cu not present; keeping prebuilt oracle/_ref" >&2; exit 0; fi
# <math_constants.h> (CUDART_PI_F): the genuine CUDA header shipped inside this image's triton wheel.
NVINC="$(python3 - <<'PY'
import os, importlib.util
s = importlib.util.find_spec("triton")
p = os.path.join(os.path.dirname(s.origin), "backends", "nvidia", "include") if s else ""
print(p if p and os.path.exists(os.path.join(p, "math_constants.h")) else "")
PY
)"
if [ -z "$NVINC" ]; then echo "build_ref: no math_constants.h in this image -> reference kernels unbuildable" >&2; exit 0; fi
mkdir -p "$HERE/_ref"
COMMON=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -include hip/hip_runtime.h -I"$SRC" -idirafter "$NVINC" -Wno-unused-value)
hipcc "${COMMON[@]}" -fno-slp-vectorize "$HERE/ref_driver.hip" -o "$HERE/_ref/libref_fma.so"
hipcc "${COMMON[@]}" -ffp-contract=off  "$HERE/ref_driver.hip" -o "$HERE/_ref/libref_nofma.so"
# Evidence of the Gaussian lowering (25 fused taps vs 0), kept next to the binaries.
for v in fma:-fno-slp-vectorize nofma:-ffp-contract=off; do
  name="${v%%:*}"; flag="${v#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S -include hip/hip_runtime.h -I"$SRC" -idirafter "$NVINC" \
        -Wno-unused-value $flag "$HERE/ref_driver.hip" -o "$HERE/_ref/ref_$name.s" 2>/dev/null
  f=$(awk '/^_ZN3cvp4cuda17gaussianFilter5x5/,/s_endpgm/' "$HERE/_ref/ref_$name.s" | grep -cE "v_fma_f32|v_fmac_f32" || true)
  m=$(awk '/^_ZN3cvp4cuda17gaussianFilter5x5/,/s_endpgm/' "$HERE/_ref/ref_$name.s" | grep -cE "v_mul_f32|v_pk_mul_f32|v_add_f32|v_pk_add_f32" || true)
  echo "gaussianFilter5x5[$name]: fma=$f mul+add=$m" | tee "$HERE/_ref/lowering_$name.txt"
  rm -f "$HERE/_ref/ref_$name.s"
done
