#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
run() { for i in 1 2; do s=$(date +%s%N); env "$@" GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "$* wall $(( (e - s) / 1000000 )) ms  $(grep 'queries packed' /tmp/e2e.err) $(grep 'gtx_set_refs done' /tmp/e2e.err)"; done; }
run A=1
run GTX_PACK_THREADS=16
run GTX_PACK_THREADS=32
run GTX_PACK_THREADS=128
run GTX_PACK_THREADS=128 GTX_READ_THREADS=16
run GTX_PACK_THREADS=64 GTX_READ_THREADS=16 GTX_PACK_BLOCK_MB=128
run GTX_PACK_THREADS=128 GTX_READ_THREADS=32 GTX_PACK_BLOCK_MB=256
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_out.txt
