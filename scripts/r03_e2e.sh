#!/bin/bash
# genomic_overlaps count -S -i from files, 100 M reads x 1 M regions: wall time from BED text / packed file, marks, exit variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03_e2e
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed; $B/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() { f=$1; shift; for i in 1 2 3; do s=$(date +%s%N); env "$@" GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "$* $(basename $f) wall $(( (e - s) / 1000000 )) ms  md5 $(md5sum < /tmp/e2e_out.txt | cut -c1-8) leaving->gone $(( e / 1000000 - $(grep -o 'leaving at epoch ms [0-9]*' /tmp/e2e.err | grep -o '[0-9]*$') )) ms"; done; cat /tmp/e2e.err; }
{ run /tmp/e2e_reads.bed A=1; run /tmp/e2e_reads.gtx A=1; run /tmp/e2e_reads.bed GTX_FULL_EXIT=1; run /tmp/e2e_reads.gtx GTX_FULL_EXIT=1; 
  s=$(date +%s%N); cat /tmp/e2e_reads.bed > /dev/null; e=$(date +%s%N); echo "cat reads.bed: $(( (e - s) / 1000000 )) ms"; } > gpurun_out/r03_e2e/e2e.txt 2>&1
grep -E "wall|cat reads" gpurun_out/r03_e2e/e2e.txt
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out.txt
