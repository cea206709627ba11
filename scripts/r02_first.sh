#!/bin/bash
# round 2, first GPU pass: the whole -m gpu suite, then the config-5 shaped bench (1 B reads x 2 M regions) with its kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_first; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
timeout -k 10 300 python3 bench.py --reads 1000000000 --refs 2000000 --cpu-sample 20000000 > $out/bench_c5_line.json 2> $out/bench_c5.err && echo "c5 bench ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c5 -- python3 bench.py --reads 1000000000 --refs 2000000 --cpu-sample 0 --steps 20 --warmup 3 > $out/bench_c5_line_under_rocprof.json 2> $out/rocprof_c5.err && echo "c5 stats ok"
cat $out/bench_c5_line.json
