#!/bin/bash
# end-to-end A/B on ONE box: variants of the command-line tool (environment knobs after the colon), alternating; wall time and how
# long the process takes to go away after its last line (GTX_TIMING prints the epoch time it leaves at)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=./ibm-cbc-genomic-tools_amd/csrc
$P/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $P/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed; $P/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
for f in ${FILES:-/tmp/e2e_reads.gtx /tmp/e2e_reads.bed}; do for rep in 1 2 3 4; do for v in ${VARIANTS:-default :GTX_LOAD_THREADS=1 :GTX_LOAD_THREADS=16}; do
  e1=A=1; [ "${v#*:}" != "$v" ] && e1=$(echo ${v#*:} | tr ',' ' ')
  s=$(date +%s%N); env $e1 GTX_TIMING=1 $P/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N)
  z=$(grep "leaving at" /tmp/e2e.err | grep -o "[0-9]*\]" | tr -d "]")
  echo "$v $(basename $f) wall $(( (e - s) / 1000000 )) ms: leaving->gone $(( ${z:-0} > 0 ? e / 1000000 - ${z:-0} : -1 )) ms, md5 $(md5sum < /tmp/e2e_out.txt | cut -c1-8)"
done; done; done
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out.txt
