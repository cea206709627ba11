#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
thr() { grep nr_throttled /sys/fs/cgroup/cpu.stat | awk '{print $2}'; }
for cfg in "6 2" "8 2" "8 4" "10 4" "12 2" "12 4" "14 2" "16 4" "64 8"; do set -- $cfg; a=$(thr); for i in 1 2; do s=$(date +%s%N); GTX_PACK_THREADS=$1 GTX_READ_THREADS=$2 GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "pack $1 read $2: wall $(( (e - s) / 1000000 )) ms  packed at $(grep 'queries packed' /tmp/e2e.err | awk '{print $2}')"; done; echo "   throttled periods: $(( $(thr) - a ))"; done
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_out.txt
