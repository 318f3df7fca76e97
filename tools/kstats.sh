#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace of one bench variant: per kernel the launches, median / minimum / mean duration of the
# second half of the launches (the first ones run while the clocks still ramp up).  Usage: tools/kstats.sh <bench args...>
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/kstats"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/s" -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
f=$(ls "$OUT"/s/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "hc::" in n:
        d[n.split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for n, v in d.items():
    v.sort()
    t = [x[1] / 1e3 for x in v[len(v) // 2:]]
    print(n[:56], "calls", len(v), "median_us", round(statistics.median(t), 1), "min_us", round(min(t), 1), "mean_us", round(sum(t) / len(t), 1))
PY
rm -rf "$OUT/s"
