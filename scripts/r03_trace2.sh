#!/bin/bash
# round 3: direct placement + tail schedule -- parity first, then the wave time line for a few schedules, then a bench line
set -e
mkdir -p gpurun_out/r03_trace2
python -m pytest tests/test_gpu_count.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py -x -q -m gpu > gpurun_out/r03_trace2/pytest.txt 2>&1 || { tail -n 30 gpurun_out/r03_trace2/pytest.txt; exit 1; }
tail -n 3 gpurun_out/r03_trace2/pytest.txt
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 56,40,28 --sched "none;-;28x8192,16x8192,8x4096;28x4096,16x4096,8x4096" --out gpurun_out/r03_trace2/wave_trace.json > gpurun_out/r03_trace2/wave_trace.txt 2>&1
python bench.py --no-e2e --cpu-sample 0 > gpurun_out/r03_trace2/bench.txt 2>&1
cut -c1-900 gpurun_out/r03_trace2/wave_trace.txt
