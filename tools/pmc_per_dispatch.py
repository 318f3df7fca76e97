#!/usr/bin/env python3
"""Per-dispatch PMC counters of the kernels whose name contains argv[2], in dispatch order (rocprofv3 --pmc CSV dir argv[1])."""
import collections
import csv
import glob
import sys

rows = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("(")[0][-40:], "grid": r.get("Grid_Size", "")})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, v in sorted(rows.items())[: int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(d, v["k"], v["grid"], {k: round(x / 1e6, 2) for k, x in v.items() if k not in ("k", "grid")})
