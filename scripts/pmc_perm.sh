#!/bin/bash
# rocprofv3 passes for the permutation_test workload of bench.py: kernel stats, then HBM fetch/write bytes and L2
# hit/miss in separate counter runs (kernel-trace only; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: scripts/pmc_perm.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- python3 bench.py --workload permutation_test --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof_${tag}_stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_${tag}_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_${tag}_write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag}_l2 -- python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_${tag}_l2.log 2>&1
