#!/usr/bin/env python3
"""Per-dispatch counters of the last complete step in a rocprofv3 --pmc run of bench.py: one line per dispatch between the
second-last and the last launch of the front kernel (its own line first), in dispatch order."""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getsize)
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = int(r["Dispatch_Id"])
    e = d.setdefault(k, {"name": r["Kernel_Name"].split("(")[0][:44], "grid": r.get("Grid_Size", "?")})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
fronts = [i for i in ids if "k_front8" in d[i]["name"]]
a, b = fronts[-2], fronts[-1]
for i in ids:
    if a <= i < b:
        e = d[i]
        print(i, e["name"], "grid", e["grid"], {c: round(v / 1e6, 3) for c, v in e.items() if c not in ("name", "grid")}, "(millions)")
