#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "owner or sorted_hint" 2>&1 | tail -2
for t in 8192 4096 2048 1024; do echo "tile $t"; GTX_SCAN_OWN_TILE=$t timeout -k 10 300 python3 scripts/bench_scan.py 2>&1 | grep "sorted-hint scan -w 500\|sorted-hint weighted scan -w 500"; done
