#!/bin/bash
mkdir -p gpurun_out/r03_share2
for pf in 0 1; do for c in 8 12 16 20 28; do echo "pf $pf cpw $c: $(GTX_PF=$pf GTX_CHUNKS_PER_WAVE=$c python scripts/share_timing.py 8 100000000 2>&1 | grep 'member 1 ' | cut -d, -f2-)"; done; done > gpurun_out/r03_share2/sweep.txt 2>&1
cat gpurun_out/r03_share2/sweep.txt
