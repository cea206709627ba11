#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/x_rehearse; mkdir -p $out
cat > $out/drive.py <<'PY'
import os, sys, time
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx, bench
from gtx import synth
os.environ["GTX_GROUP_REHEARSE"] = "1"
dev = torch.device("cuda", 0)
g = gtx.Group([0] * 4)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
per = synth.apportion(100_000_000, synth.CHROM_LEN)
owner = g.assign(per)
g.set_refs(refs, synth.n_classes())
reads = []
for m in range(4):
    ch = np.nonzero(owner == m)[0].astype(np.int64)
    reads.append(bench.make_reads_on_device(0, ch, 1000, dev, per=per[ch]))
hits = [torch.zeros(len(refs), dtype=torch.int64, device=dev) for _ in range(2)]
ptrs, ns = [r.data_ptr() for r in reads], [r.shape[0] for r in reads]
for i in range(4): g.count_device(ptrs, ns, hits[i & 1].data_ptr(), flags=gtx.READS_SORTED)
g.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): g.count_device(ptrs, ns, hits[i & 1].data_ptr(), flags=gtx.READS_SORTED)
g.sync(); torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) * 100)
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -o x -- python3 $out/drive.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
grep "ms per step" $out/log.txt
python3 - <<PY
import csv, glob
f = glob.glob('$out/t/**/x_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'count_walk' in r['Kernel_Name']]
i0 = idx[-8]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    print('%-44s q%-3s start %9.1f us  dur %7.1f us' % (r['Kernel_Name'].split('(')[0][-44:], r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
