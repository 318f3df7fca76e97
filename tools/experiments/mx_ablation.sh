#!/usr/bin/env bash
# GPU box: k_front_mx with parts left out (cudacam_amd/exp/libhipcanny_abl<N>.so, tools/build_variant.sh ... -DMX_ABL=N):
# kernel time alone (--no-pipeline) on the natural batch.  Usage: tools/experiments/mx_ablation.sh [-a "<bench args>"] <N> <N> ...
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
extra="--no-pipeline"
if [ "${1:-}" = "-a" ]; then extra="$2"; shift 2; fi
for v in "$@"; do
  export HIPCANNY_LIB="$R/cudacam_amd/exp/libhipcanny_abl$v.so"
  echo "== MX_ABL=$v $extra"
  bash "$R/tools/kstats.sh" --mx always --rotate 1 --no-host-fed $extra 2>&1 | grep front
done
