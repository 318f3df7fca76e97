#!/usr/bin/env python3
"""HBM traffic per launch of the front kernels from the PMC summary (separate rocprofv3 --pmc passes of bench.py):
2 x FETCH_SIZE + WRITE_SIZE, both in KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -- calibrated on this
code's own 4-B-per-lane row loads: the fused kernel read 531 MB of input at FETCH_SIZE = 261 508 KiB).
Usage: pmc_traffic.py <pmc_summary.txt> <bench json of the same command> > pmc_traffic.json"""
import ast, json, sys
vals = {}
for line in open(sys.argv[1]):
    if "dispatches" not in line or "{" not in line:
        continue
    name = line.split(" dispatches")[0].strip()
    d = ast.literal_eval(line[line.index("{"):])
    vals.setdefault(name, {}).update(d)
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
out = {"config": bench["config"], "metric": bench["metric"], "kernels": {}}
total = 0.0
for name, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and ("k_blur" in name or "k_nms" in name or "k_front" in name or "k_hyst" in name):
        b = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        out["kernels"][name] = {"fetch_KiB": d["FETCH_SIZE"], "write_KiB": d["WRITE_SIZE"], "hbm_bytes_per_launch": b}
        if "k_hyst" not in name:
            total += b
out["front_kernels_hbm_bytes_per_launch"] = total
print(json.dumps(out, indent=1))
