#!/bin/bash
set -e
mkdir -p gpurun_out/r03_t7
python -m pytest tests/test_gpu_count.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py tests/test_gpu_group.py -x -q -m gpu > gpurun_out/r03_t7/pytest.txt 2>&1 || { tail -n 30 gpurun_out/r03_t7/pytest.txt; exit 1; }
tail -n 2 gpurun_out/r03_t7/pytest.txt
GTX_LIB_PATH=$PWD/ibm-cbc-genomic-tools_amd/csrc/libgtx_trace.so python scripts/wave_trace.py --reads 12950000 --cpw 8,16,28 --slice-us 2 --out gpurun_out/r03_t7/wave_trace.json > gpurun_out/r03_t7/wave_trace.txt 2>&1
cut -c1-330 gpurun_out/r03_t7/wave_trace.txt
python scripts/share_timing.py 8 100000000 2>&1 | tail -n 2
python scripts/ab_count.py --rounds 3 base=ab/libgtx_base.so new=- 2>&1 | tail -n 3
python -m pytest tests/test_gpu_coverage.py tests/test_gpu_bucket.py -x -q -m gpu 2>&1 | tail -n 3
python tests/tools/bench_coverage.py 2>&1 | tail -n 4
