#!/usr/bin/env python3
"""Kernel time line from a `rocprofv3 --kernel-trace` run: the last N dispatches with queue, start (us, relative) and duration.
usage: timeline.py <dir with *_kernel_trace.csv> [N] [name filter]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40; filt = sys.argv[3] if len(sys.argv) > 3 else ""
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("gtx::", "")[:34], r.get("Queue_Id", "?")))
rows.sort()
if filt: rows = [r for r in rows if filt in r[2]]
rows = rows[-n:]
t0 = rows[0][0]
qs = sorted({r[3] for r in rows})
print("queues:", qs)
for s, e, name, q in rows:
    print("%9.1f  %7.1f us  q%-2d %s" % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), name))
