#!/bin/bash
# per-kernel times of the partition path for variant builds ab/libgtx_<name>.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x_scatter
for v in "$@"; do
  GTX_X_LIB=$GRAFT_REPO_ROOT/ab/libgtx_$v.so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x_scatter/$v -o x -- python3 scripts/bench_bucket.py > gpurun_out/x_scatter/$v.txt 2>&1 || exit 1
  f=$(find gpurun_out/x_scatter/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep 'bucket path' gpurun_out/x_scatter/$v.txt)"
  [ -n "$f" ] || { echo "no stats file"; exit 1; }
  grep "bucket_\|chunk_" "$f" | awk -F'","' '{printf "   %-60s calls %s avg %.1f us\n", substr($1,2,60), $2, $4/1000}'
done
