#!/bin/bash
for i in 1 2; do
python bench.py --no-e2e --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('async-finalize  step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
GTX_ASYNC_KERNELS=1 python bench.py --no-e2e --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('async-kernels   step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
GTX_BENCH_SYNC_FINALIZE=1 python bench.py --no-e2e --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain           step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
GTX_ASYNC_KERNELS=1 python -m pytest tests/test_gpu_count.py -x -q -m gpu -k async 2>&1 | tail -n 2
