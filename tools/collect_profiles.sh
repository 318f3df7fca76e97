#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of bench.py, summaries into gpurun_out/profiles_<tag>/.
# Usage: tools/collect_profiles.sh <tag> [bench.py arguments of the configuration to profile; none = the default command]
set -uo pipefail
TAG="${1:-r01}"
shift || true
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/profiles_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the stats pass profiles the command itself (python3 bench.py <args>); the counter passes use a short run of it
# (6 timed + 2 warm-up steps = 8 launches: two of each content of the default rotation of four)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $R/bench.py "$@" > "$OUT/bench_under_rocprof_stats.json" 2> "$OUT/stats.log"
# (the bench's host-fed leg starts tools/bin/stream_bench as a child process, which rocprofv3 traces into files of its own:
#  the bench process itself is the one with the biggest trace)
big=$(ls -S "$OUT"/stats/*/*kernel_trace.csv 2>/dev/null | head -1)
cp "${big%kernel_trace.csv}kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
case " $* " in *" --rotate 1 "*) ts=1 ;; *) ts=4 ;; esac   # the default rotation: one turn of its four contents
TRACE_STEPS=$ts python3 "$R/tools/trace_summary.py" "$OUT/stats" "$OUT/bench_under_rocprof_stats.json" > "$OUT/one_step_trace.txt" 2>&1
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo "$pass" | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$n" -- python3 $R/bench.py --steps ${PMC_STEPS:-6} --warmup ${PMC_WARMUP:-2} --no-cpu-baseline --no-host-fed "$@" > /dev/null 2> "$OUT/pmc_$n.log"
  python3 "$R/tools/pmc_summary.py" "$OUT/pmc_$n" >> "$OUT/pmc_summary.txt"
done
python3 "$R/tools/pmc_traffic.py" "$OUT/pmc_summary.txt" "$OUT/bench_under_rocprof_stats.json" > "$OUT/pmc_traffic.json"
rm -rf "$OUT"/stats "$OUT"/pmc_*/ "$OUT"/*.log
ls -la "$OUT"
