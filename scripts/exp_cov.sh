#!/bin/bash
# as exp_variants.sh, for the coverage kernels (tests/tools/bench_coverage.py)
cd "$GRAFT_REPO_ROOT/ibm-cbc-genomic-tools_amd/csrc"
for v in ${VARIANTS:-NONE}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $(echo "$v" | tr ":" "\n" | sed "s/^/-DGTX_EXP_/" | tr "\n" " ") -c gtx_kernels.hip -o gtx_kernels.o && make libgtx.so > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  for cpw in ${CPWS:-56 96}; do echo -n "$v cpw=$cpw "; (cd ../.. && GTX_CHUNKS_PER_WAVE=$cpw python tests/tools/bench_coverage.py 2>&1 | grep "coverage:\|bit-equal" | tr "\n" " "); echo; done
done
