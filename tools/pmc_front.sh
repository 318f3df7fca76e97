#!/usr/bin/env bash
# GPU box: PMC counters of k_front.  Usage: tools/pmc_front.sh "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/pmc_front"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
args="$1"; shift
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/p$i" -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline $args > /dev/null 2> "$OUT/p$i.log"
  python3 "$R/tools/pmc_summary.py" "$OUT/p$i" | grep -E "hc::" >> "$OUT/summary.txt" || tail -5 "$OUT/p$i.log"
done
rm -rf "$OUT"/p*/
cat "$OUT/summary.txt"
