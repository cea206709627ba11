#!/bin/bash
# time line of bench.py's single-rank RCCL self-test (the N > 1 step with the piece going out and back through RCCL)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/x_selftest; mkdir -p $out
GTX_BENCH_FORCE_DIST=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -o x -- python3 bench.py --steps 10 --warmup 2 --no-e2e --cpu-sample 0 > $out/line.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$out/t/**/x_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'count_walk' in r['Kernel_Name']]
i0 = idx[-4]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:idx[-1] + 8]:
    print('%-60s q%-3s start %9.1f us  dur %7.1f us' % (r['Kernel_Name'].split('(')[0][-60:], r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
cut -c1-200 $out/line.json | tail -1
