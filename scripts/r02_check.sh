#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02_check
timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "config4" 2>&1 | tail -3
timeout -k 10 600 python3 bench.py --no-e2e > gpurun_out/r02_check/bench.json 2> gpurun_out/r02_check/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r02_check/bench.json')); print(d['value'], d['roofline']['traffic'], d['cpu_baseline'])"
