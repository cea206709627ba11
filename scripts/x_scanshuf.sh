#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/x_scanshuf; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -o x -- python3 scripts/bench_scan_shuffled.py > $out/log.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$out/t/**/x_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# print the kernel sequence of the last partition-path call of each geometry: find runs
seq = [(r['Kernel_Name'].split('(')[0][-44:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'gtx::' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
agg = collections.OrderedDict()
for k, t in seq:
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += t
for k, (c, t) in agg.items(): print('%-46s calls %3d  avg %.1f us' % (k, c, t / c))
PY
grep "shuffled reads" $out/log.txt
