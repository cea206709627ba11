#!/bin/bash
mkdir -p gpurun_out/r03_trace6
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --reads 12950000 --cpw 8,16,28 --slice-us 2 --out gpurun_out/r03_trace6/wave_trace.json > gpurun_out/r03_trace6/wave_trace.txt 2>&1
cut -c1-420 gpurun_out/r03_trace6/wave_trace.txt
