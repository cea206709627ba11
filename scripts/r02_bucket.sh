#!/bin/bash
# bucket path: tests, tile / residency sweep, per-kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bucket; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_bucket.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
[ $rc -eq 0 ] || exit 1
for cfg in ${CFGS:-"2048 2" "4096 1" "4096 2"}; do
  set -- $cfg
  echo "tile $1 blocks/CU $2: $(GTX_SPLIT_TILE=$1 GTX_SPLIT_BLOCKS_PER_CU=$2 timeout -k 10 200 python3 scripts/bench_bucket.py 2>&1 | tail -1)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/bench_bucket.py > $out/bench.log 2>&1
f=$(ls -t $out/stats/*/*kernel_stats.csv | head -1); python3 - $f <<'PY'
import csv,sys
for x in csv.DictReader(open(sys.argv[1])):
    if 'gtx::' in x['Name']: print(x['Name'].split('(')[0][:60], x['Calls'], '%.4f ms' % (float(x['AverageNs'])/1e6))
PY
