#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_cov; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_coverage.py tests/test_gpu_fuzz.py tests/test_gpu_count.py -m gpu -x -q -k "cover or fuzz or merge or gaps or span_starts or batching or page_locked" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 300 python3 tests/tools/bench_coverage.py 2>&1 | tail -3
timeout -k 10 300 python3 tests/tools/bench_coverage_weighted.py 2>&1 | tail -3
