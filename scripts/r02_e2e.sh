#!/bin/bash
# where the CLI's end-to-end time goes at BASELINE config 3 size (100 M reads x 1 M regions): BED text and packed region file
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_e2e; mkdir -p $out
B=./ibm-cbc-genomic-tools_amd/csrc
N=${1:-100000000}
t() { local s=$(date +%s%N); "$@"; local rc=$?; local e=$(date +%s%N); echo "[$(( (e - s) / 1000000 )) ms] $1 $2" >&2; return $rc; }
nproc
t $B/gtx_packtool synth $N 7 /tmp/e2e_reads.bed
t $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
ls -la /tmp/e2e_reads.bed /tmp/e2e_refs.bed
t $B/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
for i in 1 2; do
s=$(date +%s%N); GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2> $out/text_$i.err > /tmp/e2e_out_text.txt; e=$(date +%s%N); echo "wall(text) $(( (e - s) / 1000000 )) ms"; grep gtx $out/text_$i.err
s=$(date +%s%N); GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.gtx 2> $out/gtx_$i.err > /tmp/e2e_out_gtx.txt; e=$(date +%s%N); echo "wall(gtx) $(( (e - s) / 1000000 )) ms"; grep gtx $out/gtx_$i.err
done
cmp /tmp/e2e_out_text.txt /tmp/e2e_out_gtx.txt && echo "same output" && md5sum /tmp/e2e_out_gtx.txt
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out_*.txt
