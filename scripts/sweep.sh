#!/bin/bash
# sweep kernel tunables on the GPU box: prints kernel_ms for each (reads per lane per step, chunks per wave)
for pf in ${PFS:-1 2 3 4}; do for cpw in ${CPWS:-24 48 96}; do
  echo -n "R=$pf cpw=$cpw "; GTX_READS_PER_LANE=$pf GTX_CHUNKS_PER_WAVE=$cpw python bench.py --cpu-sample 0 --steps 10 ${BENCH_ARGS} | grep -o '"kernel_ms[^,]*'
done; done
