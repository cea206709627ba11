#!/bin/bash
# Round-3 measurement refresh on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes (counters in their own runs, kernel-trace
# only: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Everything lands under gpurun_out/refresh_r03/; scripts/r03_collect.py copies the
# summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refresh_r03; rm -rf $out; mkdir -p $out
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
step "round 3 extras"
python3 scripts/share_timing.py 8 100000000 > $out/share_timing.txt 2>&1 && echo "share timing ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_share -- python3 scripts/share_timing.py 8 100000000 > /dev/null 2>> $out/rocprof.err && echo "share stats ok"
python3 scripts/share_timing.py 8 1000000000 > $out/share_timing_1g.txt 2>&1 && echo "share timing 1G ok"
./scripts/membench.bin 100000000 > $out/membench_100m.txt 2>&1; ./scripts/membench.bin 1000000000 > $out/membench_1g.txt 2>&1; echo "membench ok"
python3 scripts/ab_count.py --rounds 3 new=- none=-,GTX_SCHED=none > $out/ab_sched.txt 2>&1 && echo "ab ok"
GTX_BENCH_REHEARSE=1 GTX_BENCH_VERIFY=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/bench_rehearse4_line.json 2> $out/rehearse4.err && echo "rehearsal (4 members on one GPU, verified) ok"
GTX_BENCH_FORCE_DIST=1 GTX_BENCH_VERIFY=1 python3 bench.py --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/bench_selftest_line.json 2> $out/selftest.err && echo "single-rank RCCL self-test ok"
GTX_BENCH_REHEARSE=1 GTX_BENCH_BREAK_GROUP=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/bench_fallback2_line.json 2> $out/fallback2.err && echo "fallback path (group disabled on purpose, 2 ranks on one GPU) ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_pairs -- python3 scripts/bench_pairs.py > $out/bench_pairs.log 2>&1 && echo "pairs stats ok"
echo "refresh done"
