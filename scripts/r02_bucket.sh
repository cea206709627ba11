#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bucket; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_bucket.py tests/test_gpu_count.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/bench_bucket.py > $out/bench.log 2>&1; tail -2 $out/bench.log
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); cut -d, -f1-4 $f | grep -i "bucket\|finalize\|tile_sums\|gather" | cut -c1-160
