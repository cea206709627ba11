#!/bin/bash
python -m pytest tests/test_gpu_bucket.py tests/test_gpu_coverage.py -x -q -m gpu 2>&1 | tail -n 3
python scripts/bench_cov_shuffled.py 2>&1 | tail -n 4
python scripts/bench_bucket.py 2>&1 | tail -n 3
