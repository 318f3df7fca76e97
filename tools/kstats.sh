#!/usr/bin/env bash
# GPU box: rocprofv3 kernel stats of one bench variant.  Usage: tools/kstats.sh <bench args...>
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/kstats"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s" -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
f=$(ls "$OUT"/s/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("hc::",)):
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
rm -rf "$OUT/s"
