#!/bin/bash
set -e
mkdir -p gpurun_out/r03_trace5
python -m pytest tests/test_gpu_count.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py -x -q -m gpu > gpurun_out/r03_trace5/pytest.txt 2>&1 || { tail -n 30 gpurun_out/r03_trace5/pytest.txt; exit 1; }
tail -n 3 gpurun_out/r03_trace5/pytest.txt
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --cpw 56,40,28 --sched "none;-" --out gpurun_out/r03_trace5/wave_trace.json > gpurun_out/r03_trace5/wave_trace.txt 2>&1
python bench.py --no-e2e --cpu-sample 0 > gpurun_out/r03_trace5/bench.txt 2>&1
python bench.py --no-e2e --cpu-sample 0 --reads 1000000000 --refs 2000000 --steps 5 > gpurun_out/r03_trace5/bench_1g.txt 2>&1
cut -c1-330 gpurun_out/r03_trace5/wave_trace.txt; cut -c1-100 gpurun_out/r03_trace5/bench.txt;  grep -o '"kernel_ms": [0-9.]*' gpurun_out/r03_trace5/bench*.txt
