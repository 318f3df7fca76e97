#!/usr/bin/env bash
# GPU box: k_front8 / k_front8o with parts left out (cudacam_amd/exp/libhipcanny_f8abl<N>.so, tools/build_variant.sh f8abl<N>
# front8.hip -DF8_ABL=<N>): kernel time alone (--no-pipeline) and vector instructions per launch (one PMC pass), natural batch.
# Usage: tools/experiments/f8_ablation.sh <N> <N> ...
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
for v in "$@"; do
  export HIPCANNY_LIB="$R/cudacam_amd/exp/libhipcanny_f8abl$v.so"
  for mode in R O; do
    echo "== F8_ABL=$v mode $mode"
    bash "$R/tools/kstats.sh" --mode $mode --rotate 1 --no-host-fed --no-pipeline 2>&1 | grep "front"
    bash "$R/tools/pmc_front.sh" "--mode $mode --rotate 1 --no-host-fed" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" 2>&1 | grep "front" | cut -c1-200
  done
done
