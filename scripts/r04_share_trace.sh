#!/bin/bash
# kernel time line of a member's back-to-back calls (largest share of 8, 100 M x 1 M)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pipe in 1 0; do
  rm -rf gpurun_out/r04/trace_p$pipe
  GTX_GROUP_PIPELINE=$pipe rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/trace_p$pipe -o t -- python3 scripts/share_timing.py 8 100000000 > gpurun_out/r04/trace_p$pipe.txt 2>&1
  grep '^member' gpurun_out/r04/trace_p$pipe.txt
  python3 scripts/timeline.py gpurun_out/r04/trace_p$pipe 36 > gpurun_out/r04/timeline_p$pipe.txt
  rm -rf gpurun_out/r04/trace_p$pipe
done
cat gpurun_out/r04/timeline_p1.txt
