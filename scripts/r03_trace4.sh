#!/bin/bash
set -e
mkdir -p gpurun_out/r03_trace4
python -m pytest tests/test_gpu_count.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py -x -q -m gpu > gpurun_out/r03_trace4/pytest.txt 2>&1 || { tail -n 30 gpurun_out/r03_trace4/pytest.txt; exit 1; }
tail -n 3 gpurun_out/r03_trace4/pytest.txt
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 56 --sched "none;-;lin:0:512;lin:684:0;lin:684:360;lin:684:700;lin:684:1000" --out gpurun_out/r03_trace4/wave_trace.json > gpurun_out/r03_trace4/wave_trace.txt 2>&1
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 40,72 --sched "none;-" --out gpurun_out/r03_trace4/wave_trace_b.json > gpurun_out/r03_trace4/wave_trace_b.txt 2>&1
python bench.py --no-e2e --cpu-sample 0 > gpurun_out/r03_trace4/bench.txt 2>&1
cut -c1-330 gpurun_out/r03_trace4/wave_trace.txt gpurun_out/r03_trace4/wave_trace_b.txt
