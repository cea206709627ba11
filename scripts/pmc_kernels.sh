#!/bin/bash
# instruction / cycle counters of the side kernels (coverage, partition path), one counter set per run, kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_kernels; mkdir -p $out
pass() { name=$1; cmd=$2; shift 2; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 $cmd > $out/$name.log 2>&1 && echo "$name ok" || echo "$name FAILED"; }
for w in "cov tests/tools/bench_coverage.py" "bkt scripts/bench_bucket.py"; do set -- $w
  pass $1_inst $2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
  pass $1_cyc $2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  pass $1_lds $2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
done
python3 scripts/pmc_summary.py $out
