#!/bin/bash
# kernel time line of a member's back-to-back calls (largest share of 8, 100 M x 1 M)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ns in 3 1; do
  rm -rf gpurun_out/r04/trace_s$ns
  GTX_GROUP_STREAMS=$ns rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/trace_s$ns -o t -- python3 scripts/share_timing.py 8 100000000 > gpurun_out/r04/trace_s$ns.txt 2>&1
  grep '^member' gpurun_out/r04/trace_s$ns.txt
  python3 scripts/timeline.py gpurun_out/r04/trace_s$ns 45 > gpurun_out/r04/timeline_s$ns.txt
  rm -rf gpurun_out/r04/trace_s$ns
done
cat gpurun_out/r04/timeline_s3.txt
