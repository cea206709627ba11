#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bucket; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -- python3 scripts/bench_bucket.py > $out/bench_$tag.log 2>&1; grep "bucket path" $out/bench_$tag.log; python3 - <<PY
import csv,glob
f=glob.glob('$out/stats_$tag/*/*kernel_stats.csv')[0]
print('$tag', ' '.join('%s=%.3f' % (r['Name'].split('(')[0].split('::')[-1].split('<')[0], float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'bucket_' in r['Name']))
PY
}
timeout -k 10 600 python3 -m pytest tests/test_gpu_bucket.py tests/test_gpu_count.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
run base A=1
run tile4096 GTX_SPLIT_TILE=4096
run tile2048 GTX_SPLIT_TILE=2048
run blk16k GTX_COUNT_BLOCK_READS=16384
run blk64k GTX_COUNT_BLOCK_READS=65536
