#!/usr/bin/env python3
"""genomic_overlaps coverage on the GPU: 100M reads resident in HBM x 1M refs (BASELINE config 3 shape)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
from oracle import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
eng = gtx.Engine(0); eng.set_refs(refs, 24)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
cov = torch.zeros(len(refs), dtype=torch.int64, device=dev)
eng.profile(True)
for it in range(6):
    eng.coverage_device(reads.data_ptr(), n, cov.data_ptr())
eng.sync()
k = [eng.profile_last(b) for b in range(4)]
km, tm = float(np.mean([x[0] for x in k])), float(np.mean([x[1] for x in k]))
print("coverage: kernel %.3f ms, whole call %.3f ms, %.3g reads/s, %.0f GB/s algorithmic (12 B/read)" % (km, tm, n / (tm * 1e-3), 12.0 * n / (km * 1e-3) / 1e9))
ns = min(n, 20_000_000)
eng.coverage_device(reads.data_ptr(), ns, cov.data_ptr()); eng.sync()
want = orc.coverage(refs, reads[:ns].cpu().numpy(), algo=orc.SORTED_MERGE)
print("bit-equal to the oracle on %d reads:" % ns, bool(np.array_equal(cov.cpu().numpy().view(np.uint64), want)))
