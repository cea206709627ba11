#!/bin/bash
# Round-2 measurement refresh on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes (counters in their own runs, kernel-trace
# only: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Everything lands under gpurun_out/refresh_r02/; scripts/r02_collect.py copies the
# summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refresh_r02; rm -rf $out; mkdir -p $out
step() { echo "== $1"; }
step "bench lines"
python3 bench.py > $out/bench_line.json 2> $out/bench.err && echo "count ok" &&
python3 bench.py --workload scans > $out/bench_scans_line.json 2>> $out/bench.err && echo "scans ok" &&
python3 bench.py --workload permutation_test > $out/bench_perm_line.json 2>> $out/bench.err && echo "perm ok" &&
python3 bench.py --reads 1000000000 --refs 2000000 --cpu-sample 20000000 --no-e2e > $out/bench_c5_line.json 2>> $out/bench.err && echo "c5 ok"
step "kernel stats"
P="--no-e2e --cpu-sample 0 --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $P > $out/bench_line_under_rocprof.json 2> $out/rocprof.err && echo "count stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_scans -- python3 bench.py --workload scans --cpu-sample 0 --steps 20 --warmup 3 > /dev/null 2>> $out/rocprof.err && echo "scans stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_perm -- python3 bench.py --workload permutation_test --cpu-sample 0 > /dev/null 2>> $out/rocprof.err && echo "perm stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c5 -- python3 bench.py --reads 1000000000 --refs 2000000 $P > /dev/null 2>> $out/rocprof.err && echo "c5 stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_bucket -- python3 scripts/bench_bucket.py > $out/bench_bucket.log 2>&1 && echo "bucket stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cov -- python3 tests/tools/bench_coverage.py > $out/bench_cov.log 2>&1 && echo "coverage stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_scanfine -- python3 scripts/bench_scan.py > $out/bench_scan.log 2>&1 && echo "scan geometry stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_covshuf -- python3 scripts/bench_cov_shuffled.py > $out/bench_covshuf.log 2>&1 && echo "coverage (shuffled reads) stats ok"
python3 scripts/bench_scan_shuffled.py > $out/bench_scanshuf.log 2>&1 && echo "scans (shuffled reads) ok"
step "pmc"
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_$name -- "$@" > $out/pmc_$name.log 2>&1 && echo "pmc $name ok"; }
Q="--no-e2e --cpu-sample 0 --steps 3 --warmup 1"
pmc count_fetch FETCH_SIZE python3 bench.py $Q
pmc count_write WRITE_SIZE python3 bench.py $Q
pmc perm_fetch FETCH_SIZE python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0
pmc perm_write WRITE_SIZE python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0
pmc perm_l2 "TCC_HIT_sum TCC_MISS_sum" python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0
pmc bucket_fetch FETCH_SIZE python3 scripts/bench_bucket.py
pmc bucket_write WRITE_SIZE python3 scripts/bench_bucket.py
pmc cov_fetch FETCH_SIZE python3 tests/tools/bench_coverage.py
pmc cov_write WRITE_SIZE python3 tests/tools/bench_coverage.py
grep -h "bucket path\|coverage:\|coverage, \|scan -w" $out/bench_bucket.log $out/bench_cov.log $out/bench_scan.log $out/bench_covshuf.log $out/bench_scanshuf.log
echo "refresh done"
