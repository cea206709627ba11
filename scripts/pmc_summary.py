#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, cs in acc.items():
        if "gtx" not in k and "perm_" not in k: continue
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
