#!/bin/bash
# PMC passes for the bench (separate runs, kernel-trace only; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: scripts/pmc.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 $BENCH_ARGS > gpurun_out/pmc_${tag}_$name.log 2>&1; }
pass inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_WR &&
pass cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE GRBM_GUI_ACTIVE
