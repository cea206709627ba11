#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
./scripts/pmc_perm.sh r02 && python3 scripts/pmc_summary.py gpurun_out/pmc_r02_fetch && python3 scripts/pmc_summary.py gpurun_out/pmc_r02_write && python3 scripts/pmc_summary.py gpurun_out/pmc_r02_l2
