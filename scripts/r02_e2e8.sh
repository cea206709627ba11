#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=./ibm-cbc-genomic-tools_amd/csrc
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null); nproc $(nproc); affinity $(taskset -p $$ 2>/dev/null | cut -c1-80)"
grep -E "nr_throttled|throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
$B/gtx_packtool synth 100000000 7 /tmp/e2e_reads.bed; $B/gtx_packtool synthrefs 1000000 8 /tmp/e2e_refs.bed
for t in 12 16 24 32 48 64; do for i in 1 2; do s=$(date +%s%N); GTX_PACK_THREADS=$t GTX_TIMING=1 $B/genomic_overlaps count -S -i /tmp/e2e_refs.bed /tmp/e2e_reads.bed 2> /tmp/e2e.err > /tmp/e2e_out.txt; e=$(date +%s%N); echo "threads $t wall $(( (e - s) / 1000000 )) ms  $(grep 'queries packed' /tmp/e2e.err | awk '{print $2}')"; done; done
grep -E "nr_throttled|throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
rm -f /tmp/e2e_reads.bed /tmp/e2e_refs.bed /tmp/e2e_out.txt
