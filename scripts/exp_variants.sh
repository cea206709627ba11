#!/bin/bash
# timing-only experiments on the streaming kernel: rebuild libgtx.so on the GPU box with one GTX_EXP_* macro at a time
# (results are wrong by design; the macros only exist to locate where the time goes)
cd "$GRAFT_REPO_ROOT/ibm-cbc-genomic-tools_amd/csrc"
for v in ${VARIANTS:-NONE NOFLUSH NOPART NOCROSS NOB}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $(echo "$v" | tr ":" "\n" | sed "s/^/-DGTX_EXP_/" | tr "\n" " ") -c gtx_kernels.hip -o gtx_kernels.o && make libgtx.so > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  for i in 1 2; do echo -n "$v "; (cd ../.. && python bench.py --cpu-sample 0 --steps 10 ${BENCH_ARGS} 2>/dev/null | grep -o '"kernel_ms[^,]*'); done
done
