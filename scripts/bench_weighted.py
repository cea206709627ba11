#!/usr/bin/env python3
"""Kernel time of the weighted count (reads with label weights, genomic_overlaps count without -i) next to the unweighted one."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
dev = torch.device("cuda", 0)
reads = bench.make_reads_on_device(n, np.arange(24), 1000, dev)
w = torch.randint(1, 5, (n,), dtype=torch.int32, device=dev)
hits = torch.zeros(len(refs), dtype=torch.int64, device=dev)
eng = gtx.Engine(0); eng.set_refs(refs, synth.n_classes()); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.profile(True)
for name, wp in (("unweighted", None), ("weighted", w.data_ptr())):
    for _ in range(4):
        eng.count_device(reads.data_ptr(), n, hits.data_ptr(), wp, gtx.READS_SORTED)
    eng.sync()
    k = np.mean([eng.profile_last(b)[0] for b in range(3)]); t = np.mean([eng.profile_last(b)[1] for b in range(3)])
    print("%-11s kernel %.3f ms, call %.3f ms, %.3g reads/s (sum of counts %d)" % (name, k, t, n / t / 1e-3, int(hits.sum().item())))
