#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_scan; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py tests/test_gpu_cli.py tests/test_gpu_group.py -m gpu -x -q -k "scan or peaks or group or ngpu" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest.log
timeout -k 10 300 python3 scripts/bench_scan.py 2>&1 | grep scan
