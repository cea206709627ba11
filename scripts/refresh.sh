#!/bin/bash
# The round's measurement refresh on the GPU box, ONE command for every number DESIGN.md section 7 quotes: bench lines, rocprofv3 kernel
# stats of the same commands, PMC passes (counters in their own runs, kernel-trace only: MI355X_MICROARCH.md "rocprofv3 PMC slots"), the
# partition / coverage / scan / pair / permutation side benches, a group member's call at 1/8 of the reads, the bare load pattern.
#   gpurun --timeout 1200 -- 'bash scripts/refresh.sh r04 [part ...]'      parts: bench stats pmc group (default: all four), perm, scans
# Everything lands under gpurun_out/refresh_<tag>/; `python scripts/collect.py <tag>` (here, afterwards) copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r04}; shift; parts=${*:-bench stats pmc group}
out=gpurun_out/refresh_$tag; mkdir -p $out
want() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
step() { echo "== $1"; }
if want bench; then
step "bench lines"
python3 bench.py > $out/bench_line.json 2> $out/bench.err && echo "count ok" &&
python3 bench.py --workload scans > $out/bench_scans_line.json 2>> $out/bench.err && echo "scans ok" &&
python3 bench.py --workload permutation_test > $out/bench_perm_line.json 2>> $out/bench.err && echo "perm ok" &&
python3 bench.py --reads 1000000000 --refs 2000000 --cpu-sample 20000000 --no-e2e > $out/bench_c5_line.json 2>> $out/bench.err && echo "c5 ok"
fi
if want stats; then
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
python3 scripts/bench_weighted.py > $out/bench_weighted.log 2>&1 && echo "weighted count ok"
python3 tests/tools/bench_coverage_weighted.py > $out/bench_covw.log 2>&1 && echo "weighted coverage ok"
python3 scripts/bench_sort.py > $out/bench_sort.log 2>&1 && echo "sort ok"
fi
if want pmc; then
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
fi
if want scans; then
step "genomic_scans only (bench line, kernel stats, geometries): after a change to the scan kernels"
python3 bench.py --workload scans > $out/bench_scans_line.json 2>> $out/bench.err && echo "scans ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_scans -- python3 bench.py --workload scans --cpu-sample 0 --steps 20 --warmup 3 > /dev/null 2>> $out/rocprof.err && echo "scans stats ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_scanfine -- python3 scripts/bench_scan.py > $out/bench_scan.log 2>&1 && echo "scan geometry stats ok"
python3 scripts/bench_scan_shuffled.py > $out/bench_scanshuf.log 2>&1 && echo "scans (shuffled reads) ok"
fi
if want perm; then
step "permutation test only (bench line, kernel stats, PMC passes): after a change to gtx_perm.hip"
python3 bench.py --workload permutation_test > $out/bench_perm_line.json 2>> $out/bench.err && echo "perm ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_perm -- python3 bench.py --workload permutation_test --cpu-sample 0 > /dev/null 2>> $out/rocprof.err && echo "perm stats ok"
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_$name -- "$@" > $out/pmc_$name.log 2>&1 && echo "pmc $name ok"; }
pmc perm_fetch FETCH_SIZE python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0
pmc perm_write WRITE_SIZE python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0
pmc perm_l2 "TCC_HIT_sum TCC_MISS_sum" python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0
fi
if want group; then
step "a group member's call, the bare load pattern, bench.py N > 1 on one GPU"
python3 scripts/share_timing.py 8 100000000 > $out/share_timing.txt 2>&1 && echo "share timing ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_share -- python3 scripts/share_timing.py 8 100000000 > /dev/null 2>> $out/rocprof.err && echo "share stats ok"
python3 scripts/share_timing.py 8 1000000000 > $out/share_timing_1g.txt 2>&1 && echo "share timing 1G ok"
./scripts/membench.bin 100000000 > $out/membench_100m.txt 2>&1; ./scripts/membench.bin 1000000000 > $out/membench_1g.txt 2>&1; echo "membench ok"
for ns in 1 2 3; do echo "streams $ns: $(GTX_GROUP_STREAMS=$ns python3 scripts/share_timing.py 8 100000000 2>&1 | grep '^member' | tr '\n' '|')"; done > $out/share_streams.txt 2>&1 && echo "share streams ok"
GTX_BENCH_REHEARSE=1 GTX_BENCH_VERIFY=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/bench_rehearse4_line.json 2> $out/rehearse4.err && echo "rehearsal (4 members on one GPU, verified) ok"
GTX_BENCH_FORCE_DIST=1 GTX_BENCH_VERIFY=1 python3 bench.py --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/bench_selftest_line.json 2> $out/selftest.err && echo "single-rank RCCL self-test ok"
GTX_BENCH_REHEARSE=1 GTX_BENCH_BREAK_GROUP=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/bench_fallback2_line.json 2> $out/fallback2.err && echo "fallback path (group disabled on purpose, 2 ranks on one GPU) ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_pairs -- python3 scripts/bench_pairs.py > $out/bench_pairs.log 2>&1 && echo "pairs stats ok"
fi
echo "refresh done: $parts"
