#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in ${REFS:-1000000}; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmcc_$m -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --refs $m > gpurun_out/pmcc_$m.log 2>&1
  echo "refs=$m"; python3 scripts/pmc_summary.py gpurun_out/pmcc_$m | grep count_walk
  rocprofv3 --kernel-trace --pmc SQ_LEVEL_WAVES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcd_$m -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --refs $m > gpurun_out/pmcd_$m.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/pmcd_$m | grep count_walk
done
