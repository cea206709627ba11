#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
GTX_PACK_TRACE=1 GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2>&1 >/dev/null | grep -E "waited|block packed|gtx " | awk '{printf "%s | ", $0} END {print ""}' | fold -w 3000 | head -3
echo; echo "--- raw read speed of the file (dd, cat)"
s=$(date +%s%N); cat /tmp/e2e_reads.bed > /dev/null; e=$(date +%s%N); echo "cat: $(( (e - s) / 1000000 )) ms"
s=$(date +%s%N); dd if=/tmp/e2e_reads.bed of=/dev/null bs=64M 2>/dev/null; e=$(date +%s%N); echo "dd 64M: $(( (e - s) / 1000000 )) ms"
df /tmp | tail -1; mount | grep -E " /tmp | / " | head -3
for rt in 8 32; do s=$(date +%s%N); GTX_READ_THREADS=$rt $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed > /dev/null; e=$(date +%s%N); echo "read threads $rt: $(( (e - s) / 1000000 )) ms"; done
cp /tmp/e2e_reads.bed /dev/shm/e2e_reads.bed && s=$(date +%s%N); $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /dev/shm/e2e_reads.bed > /dev/null; e=$(date +%s%N); echo "from /dev/shm: $(( (e - s) / 1000000 )) ms"; rm -f /dev/shm/e2e_reads.bed
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed
