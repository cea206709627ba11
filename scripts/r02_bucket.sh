#!/bin/bash
# bucket path: tests, then the 100 M timing under a few tile / residency settings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bucket; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_bucket.py tests/test_gpu_fuzz.py tests/test_gpu_count.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -4 $out/pytest.log
[ $rc -eq 0 ] || exit 1
for cfg in "4096 2" "4096 1" "8192 1" "2048 2"; do
  set -- $cfg
  echo "tile $1 blocks/CU $2: $(GTX_SPLIT_TILE=$1 GTX_SPLIT_BLOCKS_PER_CU=$2 timeout -k 10 200 python3 scripts/bench_bucket.py 2>&1 | tail -1)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/bench_bucket.py > $out/bench.log 2>&1
f=$(ls -t $out/stats/*/*kernel_stats.csv | head -1); head -8 $f | cut -d, -f1-4
