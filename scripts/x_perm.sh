#!/bin/bash
t() { python3 bench.py --workload permutation_test --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['roofline']['apply_kernel_ms'],3))"; }
timeout -k 10 300 python -m pytest tests/test_gpu_perm.py -x -q -m gpu 2>&1 | tail -2
for c in 3 8; do GTX_PERM_CHUNKS=$c t "chunks=$c"; done
t "auto"; GTX_PERM_NO_TABLE=1 t "direct"
