#!/usr/bin/env python3
"""genomic_scans counts on the GPU: 100M reads resident in HBM, BASELINE config 4 geometries.
Prints reads/s and the algorithmic-bytes rate (12 B per read + 8 B per micro-window and per window)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
sys.path.insert(0, R)
from bench import make_reads_on_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
eng = gtx.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
w = torch.randint(1, 5, (n,), dtype=torch.int32, device=dev)
for step, size, wp, fl in ((1000, 1000, None, 0), (1000, 1000, None, 1), (25, 500, None, 0), (25, 500, None, 1), (1000, 1000, w, 0), (1000, 1000, w, 1), (25, 500, w, 1)):
    off, tot = gtx.scan_layout(synth.CHROM_LEN, step, size)
    out = torch.zeros(tot, dtype=torch.int64, device=dev)
    n_micro = int((synth.CHROM_LEN // step).sum())
    eng.profile(True)
    for it in range(6):
        eng.scan_device(reads.data_ptr(), n, synth.CHROM_LEN, step, size, out.data_ptr(), d_weights=None if wp is None else wp.data_ptr(), flags=fl)
    eng.sync()
    k = [eng.profile_last(b) for b in range(4)]
    hist_ms = float(np.mean([x[0] for x in k])); tot_ms = float(np.mean([x[1] for x in k]))
    alg = 12.0 * n + 8.0 * (n_micro + tot)
    print(("sorted-hint " if fl else "general     ") + ("weighted " if wp is not None else "") + "scan -w %d -d %d: main kernel %.3f ms, whole call %.3f ms, %.3g reads/s, %.0f GB/s algorithmic (windows=%d, sum=%d)"
          % (size, step, hist_ms, tot_ms, n / (tot_ms * 1e-3), alg / (tot_ms * 1e-3) / 1e9, tot, int(out.sum())))
