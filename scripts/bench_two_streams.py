#!/usr/bin/env python3
"""Experiment: does alternating two contexts on two streams (finalize of step i under the streaming kernel of step i+1) raise steps/s?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
n = 100_000_000; K = 40
dev = torch.device("cuda", 0)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
def run(nstreams):
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    engs = []
    for s in streams:
        e = gtx.Engine(0); e.set_refs(refs, 24); e.set_stream(s.cuda_stream); engs.append(e)
    hits = [torch.zeros(len(refs), dtype=torch.int64, device=dev) for _ in streams]
    for i in range(6): engs[i % nstreams].count_device(reads.data_ptr(), n, hits[i % nstreams].data_ptr(), None, gtx.READS_SORTED)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): engs[i % nstreams].count_device(reads.data_ptr(), n, hits[i % nstreams].data_ptr(), None, gtx.READS_SORTED)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    same = all(torch.equal(hits[0], h) for h in hits)
    print("%d stream(s): %.4f ms/step  %.3g reads/s  results equal: %s" % (nstreams, el / K * 1e3, n * K / el, same), flush=True)
    for e in engs: e.close()
run(1); run(2); run(3); run(1)
# kernel durations (HIP events around the streaming kernel on its own stream) while the two streams overlap
streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
engs = []
for s in streams:
    e = gtx.Engine(0); e.set_refs(refs, 24); e.set_stream(s.cuda_stream); e.profile(True); engs.append(e)
hits = [torch.zeros(len(refs), dtype=torch.int64, device=dev) for _ in streams]
for i in range(24): engs[i & 1].count_device(reads.data_ptr(), n, hits[i & 1].data_ptr(), None, gtx.READS_SORTED)
torch.cuda.synchronize()
for e in engs:
    print("kernel ms under overlap:", ["%.3f" % e.profile_last(b)[0] for b in range(6)], "call ms:", ["%.3f" % e.profile_last(b)[1] for b in range(3)])
