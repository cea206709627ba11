#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
$B/gtx_packtool synth 40000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
GTX_PACK_TRACE=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2>&1 >/dev/null | grep -E "pack|sink" | head -60
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed
