#!/bin/bash
# the N > 1 code of bench.py on one GPU: 2 and 4 ranks share GPU 0, reduce through gloo (not a measurement)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GTX_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
  for sc in weak strong; do
    timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 5 --warmup 2 --scaling $sc --reads 40000000 --no-e2e --cpu-sample 0 2>gpurun_out/rehearse_${n}_${sc}.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=%d %s value %.3g ms %.3f reads_per_rank %s imbalance %.3f reduce %s' % (d['n_gpus'], d['scaling'], d['value'], d['ms_per_step'], d['config']['reads_per_rank'], d['config']['imbalance'], d['config']['reduce'][:40]))" || { echo "n=$n $sc failed"; tail -5 gpurun_out/rehearse_${n}_${sc}.err; exit 1; }
  done
done
