#!/usr/bin/env python3
"""Static instruction mix of the big basic blocks of one kernel in a hipcc -S listing.
usage: isa_blocks.py file.s mangled_kernel_name [min_instrs]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
a = s.index(sys.argv[2] + ":")
b = s.index("s_endpgm", a)
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
blocks = []
name, n, c, ops = "entry", 0, Counter(), Counter()
for l in s[a:b].splitlines():
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    t = l.strip()
    if m:
        blocks.append((name, n, c, ops))
        name, n, c, ops = m.group(1), 0, Counter(), Counter()
    elif l.startswith("\t") and t and not t.startswith((".", ";")):
        op = t.split()[0]
        cat = "V" if op.startswith("v_") else "S" if op.startswith("s_") else "DS" if op.startswith("ds_") else "M"
        c[cat] += 1
        n += 1
        ops[op] += 1
blocks.append((name, n, c, ops))
print("total", sum(x[1] for x in blocks))
for bname, n, c, ops in blocks:
    if n >= minn:
        print(bname, n, dict(c))
        print("   ", ops.most_common(30))
