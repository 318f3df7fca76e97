#!/usr/bin/env bash
# GPU box: is an un-profiled bench run clocked like a profiled one?  Samples rocm-smi while bench.py runs 400 steps.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
( for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done ) > "$R/gpurun_out/clock_samples.txt" 2>&1 &
python3 "$R/bench.py" --steps 400 --warmup 8 --mx never --rotate 1 --no-cpu-baseline --no-host-fed 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fps',d['value'],'ms/step',d['ms_per_step'],'step_ms',d['step_ms'],'kernel_ms',d['roofline']['kernel_ms'])
"
wait
cat "$R/gpurun_out/clock_samples.txt" | cut -c1-200
