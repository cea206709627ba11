#!/bin/bash
set -e
mkdir -p gpurun_out/r03_trace3
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 56 --sched "none;28x8192,16x8192,8x4096" --out gpurun_out/r03_trace3/wave_trace.json > gpurun_out/r03_trace3/wave_trace.txt 2>&1
for w in 1 2 4; do GTX_WAVES_PER_BLOCK=$w GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 56,28 --sched "none" --out gpurun_out/r03_trace3/wpb$w.json > gpurun_out/r03_trace3/wpb$w.txt 2>&1; done
cut -c1-400 gpurun_out/r03_trace3/wpb*.txt
