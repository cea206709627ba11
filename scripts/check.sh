#!/bin/bash
# the -m gpu suite, then the bench's group path: rehearsals of 2 and 3 members on one GPU and the single-rank RCCL self-test, verified against the oracle
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests.txt 2>&1; rc=$?; tail -n 5 gpurun_out/r04/gpu_tests.txt; [ $rc -eq 0 ] || exit $rc
export GTX_BENCH_VERIFY=1
for n in 2 3; do
  GTX_BENCH_REHEARSE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 5 --warmup 2 --reads 20000000 --no-e2e --cpu-sample 0 2>gpurun_out/r04/rehearse_$n.err | tail -n 1 | cut -c1-300 || { tail -n 20 gpurun_out/r04/rehearse_$n.err; exit 1; }
done
GTX_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --reads 20000000 --no-e2e --cpu-sample 0 2>gpurun_out/r04/force_dist.err | tail -n 1 | cut -c1-300 || { tail -n 20 gpurun_out/r04/force_dist.err; exit 1; }
