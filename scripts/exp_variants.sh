#!/bin/bash
# timing-only experiments on the streaming kernel: rebuild libgtx.so on the GPU box with one GTX_EXP_* macro at a time and
# time the default bench with each build, on the SAME box (boxes of the pool differ by ~5 %).  The macros are put into
# gtx_kernels.hip for the duration of an experiment (#ifdef GTX_EXP_NOFLUSH ... around the piece to take out -- results are
# then wrong by design) and removed again; DESIGN.md section 7 lists what was measured this way.  NONE = the tree as it is.
cd "$GRAFT_REPO_ROOT/ibm-cbc-genomic-tools_amd/csrc"
for v in ${VARIANTS:-NONE NOFLUSH NOPART NOCROSS NOB}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $(echo "$v" | tr ":" "\n" | sed "s/^/-DGTX_EXP_/" | tr "\n" " ") -c gtx_kernels.hip -o gtx_kernels.o && make libgtx.so > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  for i in 1 2; do echo -n "$v "; (cd ../.. && python bench.py --cpu-sample 0 --steps 10 ${BENCH_ARGS} 2>/dev/null | grep -o '"kernel_ms[^,]*'); done
done
