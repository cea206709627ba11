#!/bin/bash
# end-to-end A/B on ONE box: xbuild/old and xbuild/new (binary + libgtx.so each), the new one also with parts switched off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=./xbuild/new
$P/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $P/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed; $P/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
for f in /tmp/e2e_reads.gtx /tmp/e2e_reads.bed; do for rep in 1 2 3 4; do for v in new new:GTX_LOAD_THREADS=16; do
  b=${v%%:*}; e1=A=1; [ "$v" != "$b" ] && e1=${v#*:}
  e1=$(echo $e1 | tr ',' ' ')
  s=$(date +%s%N); env $e1 GTX_TIMING=1 ./xbuild/$b/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N)
  a=$(grep "first mark" /tmp/e2e.err | grep -o "[0-9]*\]" | tr -d "]"); z=$(grep "leaving at" /tmp/e2e.err | grep -o "[0-9]*\]" | tr -d "]")
  echo "$v $(basename $f) wall $(( (e - s) / 1000000 )) ms: leaving->gone $(( ${z:-0} > 0 ? e / 1000000 - ${z:-0} : -1 )) ms"
done; done; done
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out.txt
