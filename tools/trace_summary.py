#!/usr/bin/env python3
"""Prints per-dispatch kernel durations of one bench step from a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
path = sys.argv[1]
f = sorted(glob.glob(path + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of k_front starts the last step
idx = [i for i, n in enumerate(names) if "k_front" in n]
start = idx[-1] if idx else 0
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + 40]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  {d:9.1f} us  {r["Kernel_Name"][:60]}  vgpr={r.get("VGPR_Count")} sgpr={r.get("SGPR_Count")} lds={r.get("LDS_Block_Size")} grid={r.get("Grid_Size_X")}')
