#!/bin/bash
# genomic_overlaps count -S -i from files, 100 M reads x 1 M regions: wall time from BED text and from a packed region file, and
# where it goes (GTX_TIMING marks)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed; $B/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
run() { f=$1; shift; for i in 1 2 3; do s=$(date +%s%N); env "$@" GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "$* $(basename $f) wall $(( (e - s) / 1000000 )) ms  md5 $(md5sum < /tmp/e2e_out.txt | cut -c1-8)"; done; cat /tmp/e2e.err; }
run /tmp/e2e_reads.bed A=1
run /tmp/e2e_reads.gtx A=1
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out.txt
