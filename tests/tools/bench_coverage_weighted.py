#!/usr/bin/env python3
"""Kernel time of coverage with label weights (100 M sorted reads x 1 M regions), checked against the oracle on a sample."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from oracle import orc
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
dev = torch.device("cuda", 0)
reads = bench.make_reads_on_device(n, np.arange(24), 1000, dev)
w = torch.randint(1, 5, (n,), dtype=torch.int32, device=dev)
cov = torch.zeros(len(refs), dtype=torch.int64, device=dev)
eng = gtx.Engine(0); eng.set_refs(refs, synth.n_classes()); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.profile(True)
for _ in range(4):
    eng.coverage_device(reads.data_ptr(), n, cov.data_ptr(), w.data_ptr())
eng.sync()
k = np.mean([eng.profile_last(b)[0] for b in range(3)]); t = np.mean([eng.profile_last(b)[1] for b in range(3)])
print("weighted coverage: kernel %.3f ms, whole call %.3f ms, %.3g reads/s" % (k, t, n / t / 1e-3))
ns = min(n, 10_000_000)
eng.coverage_device(reads.data_ptr(), ns, cov.data_ptr(), w.data_ptr()); eng.sync()
want = orc.coverage(refs, reads[:ns].cpu().numpy(), w[:ns].cpu().numpy(), algo=orc.SORTED_MERGE)
print("bit-equal to the oracle on %d reads: %s" % (ns, bool(np.array_equal(cov.cpu().numpy().view(np.uint64), want))))
