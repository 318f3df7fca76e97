#!/usr/bin/env python3
"""Prints per-dispatch kernel durations of one bench step from a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
path = sys.argv[1]
f = sorted(glob.glob(path + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last front-path launch but two (k_blur, or the fused k_front / k_front_o) starts a step in the steady state
idx = [i for i, n in enumerate(names) if "k_front" in n or "k_blur" in n]
start = idx[-3] if len(idx) >= 3 else (idx[-1] if idx else 0)
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + 40]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  {d:9.1f} us  {r["Kernel_Name"][:60]}  vgpr={r.get("VGPR_Count")} sgpr={r.get("SGPR_Count")} lds={r.get("LDS_Block_Size")} grid={r.get("Grid_Size_X")}')
