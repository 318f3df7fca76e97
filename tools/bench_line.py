#!/usr/bin/env python3
"""One short line per bench.py JSON line read from stdin (kernel experiments)."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(f'{d["value"]:>10.0f} f/s  step {d["ms_per_step"]:.3f} ms  {r["kernel"]} {r["kernel_ms"]:.3f} ms  frac {r["frac"]:.3f}  hyst {r["hyst_expand_ms"]:.3f}  pipe={d["config"]["pipeline"]} B={d["config"]["batch"]}', flush=True)
