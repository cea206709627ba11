#!/bin/bash
# instruction-mix PMC pass of the bench at several reference-set sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in ${REFS:-10000 1000000 4000000}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmci_$m -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --refs $m > gpurun_out/pmci_$m.log 2>&1
  echo "refs=$m"; python3 scripts/pmc_summary.py gpurun_out/pmci_$m | grep count_walk
done
