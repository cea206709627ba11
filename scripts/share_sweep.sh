#!/bin/bash
# a member's call at 1/8 of 100 M x 1 M: streams the calls of a member take in turn (GTX_GROUP_STREAMS) x span per wave
mkdir -p gpurun_out/r04
out=gpurun_out/r04/share_sweep.txt; : > $out
for ns in 1 2 3 4; do for c in 0 16 28; do
  echo "streams $ns cpw $c: $(GTX_GROUP_STREAMS=$ns GTX_CHUNKS_PER_WAVE=$c timeout -k 10 300 python scripts/share_timing.py 8 100000000 2>&1 | grep '^member' | cut -d, -f2- | tr '\n' '|')" >> $out
done; done
cat $out
