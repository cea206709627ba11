#!/bin/bash
mkdir -p gpurun_out/r03_bed12
timeout -k 10 800 python -m pytest tests/test_gpu_cli.py tests/test_gpu_perm.py::test_config5_table_10k_shuffles -x -q -m gpu > gpurun_out/r03_bed12/pytest.txt 2>&1; rc=$?
tail -n 30 gpurun_out/r03_bed12/pytest.txt
exit $rc
