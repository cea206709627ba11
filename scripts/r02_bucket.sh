#!/bin/bash
# bucket path: tests, then per-kernel times under rocprofv3 for a few settings (environment knobs), alternating on one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bucket; rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_bucket.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
[ $rc -eq 0 ] || exit 1
i=0
for rep in 1 2; do for v in ${VARIANTS:-default GTX_COUNT_BLOCK_READS=131072}; do
  i=$((i + 1)); e1=A=1; [ $v != default ] && e1=$(echo $v | tr ',' ' ')
  env $e1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s$i -- python3 scripts/bench_bucket.py ${ARGS:-} > $out/s$i.log 2>&1
  f=$(ls -t $out/s$i/*/*kernel_stats.csv | head -1)
  echo "$v: $(grep 'bucket path' $out/s$i.log) | $(python3 - $f <<'PY'
import csv,sys
r={x['Name'].split('(')[0].split('::')[-1][:22]: float(x['AverageNs'])/1e6 for x in csv.DictReader(open(sys.argv[1])) if 'gtx::bucket' in x['Name'] or 'gtx::chunk' in x['Name']}
print('  '.join('%s %.3f' % (k.replace('bucket_','').replace('_kernel',''), v) for k, v in r.items()))
PY
)"
done; done
