set -e
python -m pytest tests/test_gpu_sort.py tests/test_gpu_perm.py -m gpu -x -q > gpurun_out/s2_sort.log 2>&1 || { tail -40 gpurun_out/s2_sort.log; exit 1; }
tail -3 gpurun_out/s2_sort.log
for r in 1 2; do for L in ab/libgtx_base.so ibm-cbc-genomic-tools_amd/csrc/libgtx.so ab/libgtx_w7.so; do
  echo "== $L round $r" >> gpurun_out/s2_ab_w.txt
  GTX_LIB_PATH=$PWD/$L python scripts/bench_weighted.py >> gpurun_out/s2_ab_w.txt 2>&1
  GTX_LIB_PATH=$PWD/$L python tests/tools/bench_coverage.py >> gpurun_out/s2_ab_w.txt 2>&1
  GTX_LIB_PATH=$PWD/$L python tests/tools/bench_coverage_weighted.py >> gpurun_out/s2_ab_w.txt 2>&1
done; done
cat gpurun_out/s2_ab_w.txt
