#!/usr/bin/env python3
"""Per-dispatch timeline of one steady-state bench step from a rocprofv3 kernel-trace CSV: start (us, relative), duration,
kernel, grid size.  The step shown is the third-last launch of the front kernel that dominates the trace (the timed
configuration, not the short host-fed leg that follows it) and everything up to the next such launch."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
import os
f = max(glob.glob(path + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)   # the bench process, not the child processes of its host-fed leg
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    if "k_front" in n or "k_blur" in n:
        dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
if not dur:
    sys.exit("no front kernel in the trace")
main = dur.most_common(1)[0][0]
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"] == main]
a = idx[-3] if len(idx) >= 3 else idx[0]
b = idx[idx.index(a) + 1] + 1 if idx.index(a) + 1 < len(idx) else len(rows)
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  {d:9.1f} us  {r["Kernel_Name"].split("(")[0][:56]:56s} grid={r.get("Grid_Size_X", r.get("Grid_Size", "?"))} wg={r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))} queue={r.get("Queue_Id", "?")}')
