#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
lscpu | grep -E "Model name|MHz|Socket|NUMA node\(s\)|Thread" | tr -s ' ' | tr '\n' ';'; echo
$B/gtx_packtool synth 20000000 7 /tmp/s20.bed
C=$(cut -f1 /tmp/s20.bed | uniq | sort -u | tr '\n' ',')
for t in 1 4 16; do for i in 1 2; do s=$(date +%s%N); $B/gtx_packtool os -q -t $t -b 8000000 -c $C /tmp/s20.bed | tail -1 >/dev/null; e=$(date +%s%N); echo "packtool threads $t: $(( (e - s) / 1000000 )) ms"; done; done
GTX_PACK_TRACE=1 $B/gtx_packtool os -q -t 16 -b 8000000 -c $C /tmp/s20.bed 2>&1 | grep -E "lines counted|parsed in|waited" | head -12 | tr '\n' '|'; echo
rm -f /tmp/s20.bed
