python -m pytest tests/test_gpu_count.py tests/test_gpu_group.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py tests/test_gpu_pairs.py -x -q 2>&1 | tail -3
for rep in 1 2; do for h in 0 1; do
echo "hist32 $h: $(GTX_HIST32=$h python bench.py --steps 40 --warmup 5 --no-e2e --cpu-sample 20000000 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])')"
done; done
for h in 0 1; do echo "hist32 $h member: $(GTX_HIST32=$h python scripts/share_timing.py 8 100000000 2>&1 | grep '^member' | sed 's/ of 8.*finalize [0-9.]* ms (medians of 20, events), / /' | tr '\n' '|')"; done
