for rep in 1 2; do for mx in 512 4096; do
echo "chain max $mx: $(GTX_CHAIN_MAX_TILES=$mx python bench.py --steps 40 --warmup 5 --no-e2e --cpu-sample 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])')"
done; done
