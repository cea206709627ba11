#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_perm; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_perm.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
for mb in 100 2.75 2.0 1.4; do echo "L2 budget $mb MB"; GTX_PERM_L2_MB=$mb timeout -k 10 300 python3 bench.py --workload permutation_test --cpu-sample 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f stat %.3f apply %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['apply_kernel_ms']))"; done
