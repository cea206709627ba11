#!/bin/bash
# bench.py's N > 1 code on one GPU: rehearsal of the group path (verified), its way out (fallback), the single-rank RCCL self-test
out=gpurun_out/r03_benchn; mkdir -p $out
R="python3 -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
GTX_BENCH_REHEARSE=1 GTX_BENCH_VERIFY=1 timeout -k 10 300 $R --nproc-per-node 4 --master-port 29544 bench.py --gpus 4 --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/rehearse4.json 2> $out/rehearse4.err || { tail -5 $out/rehearse4.err; exit 1; }
GTX_BENCH_REHEARSE=1 GTX_BENCH_BREAK_GROUP=1 timeout -k 10 300 $R --nproc-per-node 2 --master-port 29545 bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/fallback2.json 2> $out/fallback2.err || { tail -5 $out/fallback2.err; exit 1; }
GTX_BENCH_FORCE_DIST=1 GTX_BENCH_VERIFY=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-e2e --cpu-sample 0 > $out/selftest.json 2> $out/selftest.err || { tail -5 $out/selftest.err; exit 1; }
for f in rehearse4 fallback2 selftest; do python3 - <<PY
import json
d = json.loads(open('$out/$f.json').read().strip().splitlines()[-1])
print('$f', d['n_gpus'], d['scaling'], '%.4f ms' % d['ms_per_step'], d['config'].get('verified', '-')[:40], '|', d['config']['parallelism'][-120:])
PY
done
