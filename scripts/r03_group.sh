#!/bin/bash
mkdir -p gpurun_out/r03_group
timeout -k 10 800 python -m pytest tests/test_gpu_group.py -x -q -m gpu > gpurun_out/r03_group/pytest.txt 2>&1; rc=$?
tail -n 40 gpurun_out/r03_group/pytest.txt
exit $rc
