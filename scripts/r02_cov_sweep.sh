#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cpw in 16 24 32 48 56 64 96; do echo "cpw $cpw"; GTX_CHUNKS_PER_WAVE=$cpw timeout -k 10 120 python3 tests/tools/bench_coverage.py 2>&1 | grep "coverage:"; done
