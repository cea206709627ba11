#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
timeout -k 10 600 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_group.py -m gpu -x -q 2>&1 | tail -3
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed; $B/gtx_packtool pack /tmp/e2e_reads.bed /tmp/e2e_reads.gtx
run() { f=$1; shift; for i in 1 2 3; do s=$(date +%s%N); env "$@" GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "$* $(basename $f) wall $(( (e - s) / 1000000 )) ms  $(grep 'queries packed' /tmp/e2e.err) $(grep 'device ready' /tmp/e2e.err) md5 $(md5sum < /tmp/e2e_out.txt | cut -c1-8)"; done; }
run /tmp/e2e_reads.bed A=1
run /tmp/e2e_reads.bed GTX_NO_PINNED_BATCHES=1
run /tmp/e2e_reads.gtx A=1
run /tmp/e2e_reads.gtx GTX_NO_PINNED_BATCHES=1
GTX_PACK_TRACE=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2>&1 >/dev/null | grep "block packed" | head -12 | tr '\n' ' '
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_reads.gtx /tmp/e2e_out.txt
