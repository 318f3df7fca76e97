#!/usr/bin/env python3
"""HBM traffic per launch of the front kernels from the PMC summary (separate rocprofv3 --pmc passes of bench.py):
2 x FETCH_SIZE + WRITE_SIZE, both in KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -- calibrated on this
code's own 4-B-per-lane row loads: the fused kernel read 531 MB of input at FETCH_SIZE = 261 508 KiB).
Usage: pmc_traffic.py <pmc_summary.txt> <bench json of the same command> > pmc_traffic.json"""
import ast, json, sys
import re
vals, disp = {}, {}
for line in open(sys.argv[1]):
    if "dispatches" not in line or "{" not in line:
        continue
    name = line.split(" dispatches")[0].strip()
    d = ast.literal_eval(line[line.index("{"):])
    vals.setdefault(name, {}).update(d)
    disp[name] = int(line.split(" dispatches")[1].split()[0])


def base(name):
    """k_front8<IN, PROV, HALF, ONE>: the one-wave and four-wave forms (HC_OPT_FRONT_WPB, chosen at run time) are one kernel here."""
    m = re.match(r"(.*k_front8<[^,>]+,[^,>]+,[^,>]+)(?:,[^>]*)?>?", name)
    return m.group(1) + ">" if m else name
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
out = {"config": bench["config"], "metric": bench["metric"], "kernels": {}}
groups = {}
for name, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and ("k_blur" in name or "k_nms" in name or "k_front" in name or "k_hyst" in name):
        b = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        out["kernels"][name] = {"fetch_KiB": d["FETCH_SIZE"], "write_KiB": d["WRITE_SIZE"], "hbm_bytes_per_launch": b, "dispatches": disp.get(name)}
        if "k_hyst" not in name:
            g = groups.setdefault(base(name), [0.0, 0])
            g[0] += b * disp.get(name, 1)
            g[1] += disp.get(name, 1)
# per launch of the front path: the forms of one kernel averaged by their dispatch counts, different kernels (k_blur + k_nms) added
out["front_kernels_hbm_bytes_per_launch"] = sum(t / n for t, n in groups.values() if n)
print(json.dumps(out, indent=1))
