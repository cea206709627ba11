#!/usr/bin/env python3
"""Kernel time of the sorted count with and without GTX_CHECK_SORTED (order verified on the device)."""
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
hits = torch.zeros(len(refs), dtype=torch.int64, device=dev)
eng = gtx.Engine(0); eng.set_refs(refs, synth.n_classes()); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.profile(True)
for name, fl in (("sorted hint", gtx.READS_SORTED), ("sorted hint + order check", gtx.READS_SORTED | gtx.CHECK_SORTED)):
    for _ in range(4):
        eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, fl)
    eng.sync()
    k = np.mean([eng.profile_last(b)[0] for b in range(3)])
    print("%-26s kernel %.3f ms  first_unsorted=%s" % (name, k, eng.last_info()["first_unsorted"]))
