#!/usr/bin/env python3
"""genomic_scans counts with the reads in random order: 100 M reads, the two geometries of bench_scan.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
reads = reads[torch.randperm(n, device=dev)]
eng = gtx.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
for step, size in ((1000, 1000), (25, 500)):
    off, tot = gtx.scan_layout(synth.CHROM_LEN, step, size)
    out = torch.zeros(tot, dtype=torch.int64, device=dev)
    for name, flags in (("partition path (GTX_READS_UNSORTED)", gtx.READS_UNSORTED), ("general kernels (no hint)", 0)):
        eng.profile(True)
        for it in range(4): eng.scan_device(reads.data_ptr(), n, synth.CHROM_LEN, step, size, out.data_ptr(), flags=flags)
        eng.sync()
        print("shuffled reads, scan -w %d -d %d, %s: whole call %.2f ms (sum=%d)" % (size, step, name, np.mean([eng.profile_last(b)[1] for b in range(2)]), int(out.sum())))
