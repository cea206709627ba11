#!/bin/bash
mkdir -p gpurun_out/r03_text
timeout -k 10 700 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -k "text_on_device" > gpurun_out/r03_text/pytest_text.txt 2>&1; rc=$?
tail -n 40 gpurun_out/r03_text/pytest_text.txt
exit $rc
