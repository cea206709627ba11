#!/bin/bash
python scripts/share_timing.py 8 100000000 2>&1 | tail -n 2
python scripts/share_timing.py 8 1000000000 2>&1 | tail -n 2
python -m pytest tests/test_gpu_count.py tests/test_gpu_group.py -x -q -m gpu 2>&1 | tail -n 2
