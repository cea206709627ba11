#!/bin/bash
# the host-side tests that go through the command-line tools (+ group API), then the end-to-end timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_tests; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_group.py tests/test_gpu_errors.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit 1
VARIANTS=default timeout -k 10 600 ./scripts/r02_e2e_ab.sh
