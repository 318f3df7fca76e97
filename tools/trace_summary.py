#!/usr/bin/env python3
"""Per-dispatch timeline of steady-state bench steps from a rocprofv3 kernel-trace CSV: start (us, relative), duration,
kernel, grid size.  First the duration of EVERY launch of the front kernel that dominates the trace, in launch order (the
warm-up, the timed steps, then the bench's untimed legs: same-batch, host-fed, output check); then the timeline of
TRACE_STEPS consecutive steps (default 1; 4 = one turn of the default content rotation) beginning a third of the way into
those launches, i.e. inside the timed region of the default command and of the profile commands.  With the bench's JSON
line as second argument: the mean duration of the timed launches beside the figure bench.py measured with HIP events."""
import collections
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
nsteps = int(os.environ.get("TRACE_STEPS", "1"))
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


def base(name):
    """k_front8<IN, PROV, HALF, ONE>: the one-wave and four-wave forms (chosen at run time) count as one kernel."""
    m = re.match(r"(.*k_front8<[^,>]+,[^,>]+,[^,>]+)", name)
    return m.group(1) if m else name


idx = [i for i, r in enumerate(rows) if base(r["Kernel_Name"]) == base(main)]
print("front kernel launches in order, us: " + " ".join(f'{(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3:.0f}' for i in idx))
if len(sys.argv) > 2:   # the bench's JSON line: mean over its timed launches (kernel_stats.csv averages every launch of the process, untimed legs included)
    import json
    try:
        j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        w, s = int(j["warmup"]), int(j["steps"])
        timed = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e6 for i in idx[w:w + s]]
        print(f'timed launches {w} .. {w + s - 1}: mean {sum(timed) / len(timed):.4f} ms (bench.py reports kernel_ms {j["roofline"]["kernel_ms"]} from HIP events)')
    except Exception as e:   # a summary tool: say so and carry on
        print("no bench line to compare with:", e)
k = len(idx) // 3
a = idx[k]
b = idx[k + nsteps] if k + nsteps < len(idx) else len(rows)
print(f"timeline of launches {k} .. {k + nsteps - 1} of {len(idx)}:")
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  {d:9.1f} us  {r["Kernel_Name"].split("(")[0][:56]:56s} grid={r.get("Grid_Size_X", r.get("Grid_Size", "?"))} wg={r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))} queue={r.get("Queue_Id", "?")}')
