#!/bin/bash
mkdir -p gpurun_out/r03_ab2
python scripts/ab_count.py --rounds 3 base=ab/libgtx_base.so none=-,GTX_SCHED=none t512=-,GTX_SCHED=lin:0:512 t384=-,GTX_SCHED=lin:0:384 t640=-,GTX_SCHED=lin:0:640 t512m4=-,GTX_SCHED=lin:0:512:4 t400m4=-,GTX_SCHED=lin:0:400:4 h256t512=-,GTX_SCHED=lin:256:512 c48t=-,GTX_CHUNKS_PER_WAVE=48,GTX_SCHED=lin:0:560 c64t=-,GTX_CHUNKS_PER_WAVE=64,GTX_SCHED=lin:0:460 > gpurun_out/r03_ab2/ab.txt 2>&1
tail -n 12 gpurun_out/r03_ab2/ab.txt
