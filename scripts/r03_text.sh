#!/bin/bash
# device-side tokenizer: the CLI suite with the text path forced on (every streamed BED file that qualifies goes through it), then e2e timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03_text
GTX_TEXT_ON_DEVICE=1 timeout -k 10 700 python -m pytest tests/test_gpu_cli.py tests/test_gpu_group.py::test_cli_ngpu -x -q -m gpu > gpurun_out/r03_text/pytest_forced.txt 2>&1; rc=$?
tail -n 25 gpurun_out/r03_text/pytest_forced.txt
[ $rc -eq 0 ] || exit $rc
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
run() { f=$1; shift; for i in 1 2 3; do s=$(date +%s%N); env "$@" GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed $f 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "$* $(basename $f) wall $(( (e - s) / 1000000 )) ms md5 $(md5sum < /tmp/e2e_out.txt | cut -c1-8) leaving->gone $(( e / 1000000 - $(grep -o 'leaving at epoch ms [0-9]*' /tmp/e2e.err | grep -o '[0-9]*$') )) ms main $(grep 'output written' /tmp/e2e.err | grep -o '[0-9.]* s')"; done; cat /tmp/e2e.err; }
{ run /tmp/e2e_reads.bed GTX_TEXT_ON_DEVICE=0; run /tmp/e2e_reads.bed A=1; } > gpurun_out/r03_text/e2e.txt 2>&1
grep -E "wall" gpurun_out/r03_text/e2e.txt
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_out.txt
