#!/bin/bash
# A/B on one box: two builds of libgtx.so (csrc/libgtx.so and $1), the bucket path's kernels under rocprofv3, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_ab; rm -rf $out; mkdir -p $out
for rep in 1 2; do for v in A B; do
  unset GTX_X_LIB; [ $v = B ] && export GTX_X_LIB=$PWD/$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v$rep -- python3 scripts/bench_bucket.py > $out/$v$rep.log 2>&1
  f=$(ls -t $out/$v$rep/*/*kernel_stats.csv | head -1)
  echo "$v$rep $(tail -1 $out/$v$rep.log) | $(python3 - $f <<'PY'
import csv,sys
r={x['Name'].split('(')[0].split('::')[-1][:22]: float(x['AverageNs'])/1e6 for x in csv.DictReader(open(sys.argv[1])) if 'gtx::bucket' in x['Name'] or 'gtx::chunk' in x['Name']}
print('  '.join('%s %.3f' % (k.replace('bucket_','').replace('_kernel',''), v) for k, v in r.items()))
PY
)"
done; done
