#!/bin/bash
# as exp_variants.sh, for the weighted count kernel (scripts/bench_weighted.py); GTX_EXP_WLB=<waves per SIMD> -> launch bounds + SGPR cap
cd "$GRAFT_REPO_ROOT/ibm-cbc-genomic-tools_amd/csrc"
for v in ${VARIANTS:-NONE}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $(echo "$v" | tr ":" "\n" | sed "s/^/-DGTX_EXP_/" | tr "\n" " ") -c gtx_kernels.hip -o gtx_kernels.o && make libgtx.so > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  for i in 1 2; do echo -n "$v "; (cd ../.. && python scripts/bench_weighted.py 2>&1 | grep "^weighted"); done
done
