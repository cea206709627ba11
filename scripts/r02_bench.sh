#!/bin/bash
# default bench line (with the e2e objects) + scans + perm lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_bench; mkdir -p $out
s=$(date +%s); timeout -k 10 600 python3 bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"; cat $out/bench_line.json; tail -n 5 $out/bench.err
