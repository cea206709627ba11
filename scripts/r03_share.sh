#!/bin/bash
# a member's local work at 1/8 of 100 M (+ rocprof kernel stats), bench rehearsals of the group path, RCCL self-test
mkdir -p gpurun_out/r03_share
python scripts/share_timing.py 8 100000000 > gpurun_out/r03_share/share.txt 2>&1; tail -n 3 gpurun_out/r03_share/share.txt
for c in 8 16 24 32; do echo "cpw $c"; GTX_CHUNKS_PER_WAVE=$c python scripts/share_timing.py 8 100000000 2>&1 | tail -n 2; done > gpurun_out/r03_share/share_cpw.txt 2>&1; cat gpurun_out/r03_share/share_cpw.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_share/prof -o share -- python3 scripts/share_timing.py 8 100000000 > gpurun_out/r03_share/rocprof.txt 2>&1
find gpurun_out/r03_share/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_share/share_kernel_stats.csv
head -n 8 gpurun_out/r03_share/share_kernel_stats.csv | cut -c1-200
export GTX_BENCH_REHEARSE=1 GTX_BENCH_VERIFY=1
for n in 2 3; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 5 --warmup 2 --reads 20000000 --no-e2e --cpu-sample 0 2>gpurun_out/r03_share/rehearse_$n.err | tail -n 1 | cut -c1-400 || { tail -n 20 gpurun_out/r03_share/rehearse_$n.err; exit 1; }
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 3 --warmup 1 --workload scans --reads 10000000 --no-e2e --cpu-sample 0 2>gpurun_out/r03_share/rehearse_scans_$n.err | tail -n 1 | cut -c1-300 || { tail -n 20 gpurun_out/r03_share/rehearse_scans_$n.err; exit 1; }
done
unset GTX_BENCH_REHEARSE
GTX_BENCH_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --reads 20000000 --no-e2e --cpu-sample 0 2>gpurun_out/r03_share/force_dist.err | tail -n 1 | cut -c1-300
