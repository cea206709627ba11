#!/bin/bash
# permutation_test kernels: tests, bench line (10 k shuffles), and the three counter passes of perm_stat_kernel; GTX_PERM_ROWS32=1 for comparison
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_perm; rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_perm.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit 1
for v in 16 32; do
  [ $v = 32 ] && export GTX_PERM_ROWS32=1
  python3 bench.py --workload permutation_test --cpu-sample 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows$v: ms_per_step %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline'].get('kernel_ms', 0)))"
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${v}_$n -- python3 bench.py --workload permutation_test --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
    f=$(ls $out/pmc_${v}_$n/*/*counter_collection.csv | head -1)
    python3 - $f "$v $c" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'perm_stat_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: (len(v), sum(v)/len(v)) for k,v in acc.items()})
PY
  done
done
