#!/bin/bash
# round 3, first look: bare load pattern (membench) at 100 M and 1 G reads, wave time line of the streaming kernel, a bench line
set -e
mkdir -p gpurun_out/r03_trace
./scripts/membench.bin 100000000 > gpurun_out/r03_trace/membench_100m.txt 2>&1
./scripts/membench.bin 1000000000 > gpurun_out/r03_trace/membench_1g.txt 2>&1
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 56,28,112 --out gpurun_out/r03_trace/wave_trace.json > gpurun_out/r03_trace/wave_trace.txt 2>&1
python bench.py --no-e2e --cpu-sample 0 > gpurun_out/r03_trace/bench.txt 2>&1
tail -n 40 gpurun_out/r03_trace/membench_100m.txt gpurun_out/r03_trace/membench_1g.txt
