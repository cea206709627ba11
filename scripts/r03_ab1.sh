#!/bin/bash
mkdir -p gpurun_out/r03_ab1
python scripts/ab_count.py --rounds 3 base=ab/libgtx_base.so new_none=-,GTX_SCHED=none new_ramp=- new_tail=-,GTX_SCHED=lin:0:512 pf_none=-,GTX_PF=1,GTX_SCHED=none pf_ramp=-,GTX_PF=1 new40=-,GTX_CHUNKS_PER_WAVE=40 new72=-,GTX_CHUNKS_PER_WAVE=72 > gpurun_out/r03_ab1/ab.txt 2>&1
tail -n 12 gpurun_out/r03_ab1/ab.txt
