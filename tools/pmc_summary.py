#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter CSVs per kernel (sum over dispatches / number of dispatches)."""
import csv, glob, sys, collections
path = sys.argv[1]
for f in sorted(glob.glob(path + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    for k in acc:
        n = len(cnt[k])
        print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
