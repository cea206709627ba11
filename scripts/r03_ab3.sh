#!/bin/bash
mkdir -p gpurun_out/r03_ab3
python scripts/ab_count.py --rounds 6 base=ab/libgtx_base.so none=-,GTX_SCHED=none t512=-,GTX_SCHED=lin:0:512 t640=-,GTX_SCHED=lin:0:640 t800=-,GTX_SCHED=lin:0:800 c48t=-,GTX_CHUNKS_PER_WAVE=48,GTX_SCHED=lin:0:560 > gpurun_out/r03_ab3/ab.txt 2>&1
tail -n 8 gpurun_out/r03_ab3/ab.txt
./scripts/membench.bin 100000000 2>&1 | head -8
