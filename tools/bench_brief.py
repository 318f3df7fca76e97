#!/usr/bin/env python3
"""Runs bench.py with the given extra arguments and prints one compact line."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("bench failed:", out.stderr[-800:])
    sys.exit(1)
d = json.loads(line[-1])
r = d["roofline"]
print(" ".join(sys.argv[1:]), "| fps", d["value"], "ms/step", d["ms_per_step"], "front_ms", r["kernel_ms"], "hyst_ms", r["hyst_expand_ms"],
      "frac", r["frac"], d["hysteresis"], "pipeline", d["config"].get("pipeline"))
