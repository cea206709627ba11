#!/bin/bash
# group tests + two-rank gloo rehearsal of bench.py on one GPU (both ranks on device 0) with parity verification
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_group; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_group.py tests/test_gpu_count.py tests/test_gpu_scan.py tests/test_gpu_cli.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
GTX_BENCH_REHEARSE=1 GTX_BENCH_VERIFY=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --reads 5000000 --refs 100000 > $out/rehearse_weak.json 2> $out/rehearse_weak.err; echo "rehearse weak rc=$?"; cat $out/rehearse_weak.json
GTX_BENCH_REHEARSE=1 GTX_BENCH_VERIFY=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 3 --steps 3 --warmup 1 --reads 6000000 --refs 100000 --scaling strong > $out/rehearse_strong.json 2> $out/rehearse_strong.err; echo "rehearse strong rc=$?"; cat $out/rehearse_strong.json
GTX_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e > $out/force_dist.json 2> $out/force_dist.err; echo "force dist rc=$?"; cat $out/force_dist.json
for f in $out/*.err; do tail -n 3 $f; done
