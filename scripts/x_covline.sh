#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/x_covline; mkdir -p $out
for l in -1 0 4 8; do
  if [ $l = -1 ]; then unset GTX_SPLIT_LINE; else export GTX_SPLIT_LINE=$l; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s$l -o x -- python3 scripts/bench_cov_shuffled.py > $out/s$l.txt 2>&1 || exit 1
  echo "== line $l: $(grep 'partition path' $out/s$l.txt)"
  python3 - <<PY
import csv, glob
f = glob.glob('$out/s$l/**/x_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bucket_' in r['Name']: print('   %-44s calls %s avg %.1f us' % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
