#!/bin/bash
# kernel time of the streaming kernel against the number of reference regions (same 100 M reads)
for refs in ${REFS:-10000 100000 300000 1000000 2000000}; do
  echo -n "refs=$refs "; python bench.py --cpu-sample 0 --steps 10 --refs $refs ${BENCH_ARGS} 2>/dev/null | grep -o '"kernel_ms[^,]*'
done
