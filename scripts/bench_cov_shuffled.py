"""coverage of shuffled reads: the partition path against the streaming kernel (100 M reads x 1 M regions)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
reads = reads[torch.randperm(n, device=dev)]
eng = gtx.Engine(0); eng.set_refs(refs, 24); eng.set_stream(torch.cuda.current_stream().cuda_stream)
cov = torch.zeros(len(refs), dtype=torch.int64, device=dev); eng.profile(True)
for name, flags, reps in (("partition path (GTX_READS_UNSORTED)", gtx.READS_UNSORTED, 4), ("streaming kernel (no hint)", 0, 2)):
    for it in range(reps): eng.coverage_device(reads.data_ptr(), n, cov.data_ptr(), None, flags)
    eng.sync(); print("coverage, %d shuffled reads, %s: whole call %.2f ms" % (n, name, np.mean([eng.profile_last(b)[1] for b in range(2)])))
