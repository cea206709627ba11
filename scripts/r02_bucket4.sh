#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bucket; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -- python3 scripts/bench_bucket.py > $out/bench_$tag.log 2>&1; grep "bucket path" $out/bench_$tag.log; python3 - <<PY
import csv,glob,os
f=sorted(glob.glob('$out/stats_$tag/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
print('$tag', ' '.join('%s=%.3f' % (r['Name'].split('(')[0].split('::')[-1].split('<')[0], float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'bucket_' in r['Name']))
PY
}
timeout -k 10 600 python3 -m pytest tests/test_gpu_bucket.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
run t4096_1024 A=1
run t4096_512 GTX_SPLIT_THREADS=512
run t2048_512 GTX_SPLIT_THREADS=512 GTX_SPLIT_TILE=2048
run t8192_1024 GTX_SPLIT_TILE=8192
