#!/usr/bin/env python3
"""Start/end (us) of the hipcanny kernels of the last steps in a rocprofv3 kernel-trace directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_front" in r["Kernel_Name"] or "k_blur" in r["Kernel_Name"]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = int(rows[idx[-nsteps]]["Start_Timestamp"])
for r in rows[idx[-nsteps]:]:
    if "rocclr" in r["Kernel_Name"]:
        continue
    n = r["Kernel_Name"].split("(")[0][-16:]
    print("%9.1f %9.1f  %-16s q=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, n, r["Queue_Id"]))
