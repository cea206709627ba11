#!/bin/bash
# End-of-round measurement refresh on the GPU box: bench lines (count, scans, permutation_test), rocprofv3 kernel stats
# of the default bench command, PMC passes (scripts/pmc.sh).  Everything lands under gpurun_out/refresh_<tag>/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r01}; out=gpurun_out/refresh_$tag; mkdir -p $out
python3 bench.py > $out/bench_line.json 2> $out/bench.err && echo "bench ok" &&
python3 bench.py --two-streams --cpu-sample 0 > $out/bench_line_two_streams.json 2>> $out/bench.err && echo "two streams ok" &&
python3 bench.py --workload scans > $out/bench_scans_line.json 2>> $out/bench.err && echo "scans ok" &&
python3 bench.py --workload permutation_test > $out/bench_perm_line.json 2>> $out/bench.err && echo "perm ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 > $out/bench_line_under_rocprof.json 2> $out/rocprof.err && echo "stats ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_scans -- python3 bench.py --workload scans --steps 20 --warmup 3 --cpu-sample 0 > $out/bench_scans_line_under_rocprof.json 2>> $out/rocprof.err && echo "scans stats ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_perm -- python3 bench.py --workload permutation_test --cpu-sample 0 > $out/bench_perm_line_under_rocprof.json 2>> $out/rocprof.err && echo "perm stats ok" &&
./scripts/pmc.sh $tag && python3 scripts/pmc_summary.py gpurun_out > $out/pmc_summary.txt && echo "pmc ok"
