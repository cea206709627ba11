#!/bin/bash
# run a subset of the GPU tests: scripts/r02_tests.sh <pytest args>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02_tests
timeout -k 10 900 python3 -m pytest "$@" > gpurun_out/r02_tests/pytest.log 2>&1; rc=$?
tail -40 gpurun_out/r02_tests/pytest.log
exit $rc
