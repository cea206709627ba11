#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03_e2e
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed; $B/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
run() { f=$1; shift; for i in 1 2 3 4; do s=$(date +%s%N); env "$@" GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "$* $(basename $f) wall $(( (e - s) / 1000000 )) ms leaving->gone $(( e / 1000000 - $(grep -o 'leaving at epoch ms [0-9]*' /tmp/e2e.err | grep -o '[0-9]*$') )) ms main $(grep 'output written' /tmp/e2e.err | grep -o '[0-9.]* s')"; done; }
{ run /tmp/e2e_reads.bed GTX_EXIT_EXPERIMENT=1; run /tmp/e2e_reads.bed GTX_EXIT_EXPERIMENT=2; run /tmp/e2e_reads.bed GTX_EXIT_EXPERIMENT=3; run /tmp/e2e_reads.bed GTX_PACK_THREADS=8; run /tmp/e2e_reads.bed A=1; } > gpurun_out/r03_e2e/e2e3.txt 2>&1
cat gpurun_out/r03_e2e/e2e3.txt
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out.txt
