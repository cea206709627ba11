#!/bin/bash
# the multi-interval count tests (and the rest of the CLI / class-API tests they share code with)
mkdir -p gpurun_out/r03_bed12
timeout -k 10 900 python -m pytest tests/test_gpu_cli.py tests/test_class_api.py -x -q -m gpu > gpurun_out/r03_bed12/pytest.log 2>&1; rc=$?
tail -n 40 gpurun_out/r03_bed12/pytest.log
exit $rc
