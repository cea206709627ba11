#!/bin/bash
# partition path: per-kernel times and WRITE_SIZE / FETCH_SIZE of the scatter pass for GTX_SPLIT_LINE = $@
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/x_line; mkdir -p $out
for l in "$@"; do
  export GTX_SPLIT_LINE=$l
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s$l -o x -- python3 scripts/bench_bucket.py > $out/s$l.txt 2>&1 || exit 1
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p${l}_$c -o x -- python3 scripts/bench_bucket.py > $out/p${l}_$c.txt 2>&1 || exit 1
  done
  python3 - <<PY
import csv, glob
f = glob.glob('$out/s$l/**/x_kernel_stats.csv', recursive=True)[0]
print('== line $l:', open('$out/s$l.txt').read().strip().splitlines()[-1])
for r in csv.DictReader(open(f)):
    if 'bucket_' in r['Name'] or 'chunk_' in r['Name']:
        print('   %-36s calls %s avg %.1f us' % (r['Name'].split('(')[0][-36:], r['Calls'], float(r['AverageNs']) / 1e3))
for c in ('WRITE_SIZE', 'FETCH_SIZE'):
    f = glob.glob('$out/p${l}_%s/**/x_counter_collection.csv' % c, recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'bucket_scatter' in r['Kernel_Name'] and r['Counter_Name'] == c]
    print('   scatter %s: mean %.0f KB over %d dispatches' % (c, sum(v) / max(len(v), 1), len(v)))
PY
done
