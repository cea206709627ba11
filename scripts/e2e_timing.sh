#!/bin/bash
# where does the CLI's end-to-end time go? (GTX_TIMING marks)
cd "$GRAFT_REPO_ROOT"
python3 - <<PY
import sys, os
sys.path.insert(0, "ibm-cbc-genomic-tools_amd")
import numpy as np, pandas as pd
from gtx import synth
names = np.array(synth.CHROM_NAMES)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000); reads = synth.genome_intervals(20_000_000, 44, 50, 51)
pd.DataFrame({"c": names[refs[:,0]], "s": refs[:,1]-1, "e": refs[:,2], "l": ["g%d"%i for i in range(len(refs))]}).to_csv("/tmp/t_refs.bed", sep="\t", header=False, index=False)
pd.DataFrame({"c": names[reads[:,0]], "s": reads[:,1]-1, "e": reads[:,2]}).to_csv("/tmp/t_reads.bed", sep="\t", header=False, index=False)
PY
GTX_TIMING=1 ./ibm-cbc-genomic-tools_amd/csrc/genomic_overlaps count -S -i /tmp/t_refs.bed /tmp/t_reads.bed 2>&1 >/dev/null | grep gtx
# whole-process wall time, output to a file (three runs)
for i in 1 2 3; do s=$(date +%s%N); ./ibm-cbc-genomic-tools_amd/csrc/genomic_overlaps count -S -i /tmp/t_refs.bed /tmp/t_reads.bed > /tmp/t_out.txt; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms, $(wc -l < /tmp/t_out.txt) lines"; done
# the same query set as a packed region file (tokenised once)
s=$(date +%s%N); ./ibm-cbc-genomic-tools_amd/csrc/gtx_packtool pack /tmp/t_reads.bed /tmp/t_reads.gtx; e=$(date +%s%N); echo "pack: $(( (e - s) / 1000000 )) ms, $(stat -c %s /tmp/t_reads.gtx) bytes"
GTX_TIMING=1 ./ibm-cbc-genomic-tools_amd/csrc/genomic_overlaps count -S -i /tmp/t_refs.bed /tmp/t_reads.gtx 2>&1 >/tmp/t_out2.txt | grep "gtx " | tail -4
cmp /tmp/t_out.txt /tmp/t_out2.txt && echo "same output"
for i in 1 2 3; do s=$(date +%s%N); ./ibm-cbc-genomic-tools_amd/csrc/genomic_overlaps count -S -i /tmp/t_refs.bed /tmp/t_reads.gtx > /tmp/t_out2.txt; e=$(date +%s%N); echo "wall (packed input) $(( (e - s) / 1000000 )) ms"; done
