#!/usr/bin/env python3
"""100M shuffled reads x 1M refs through the bucket path (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100_000_000
weighted = "--weights" in sys.argv
dev = torch.device("cuda", 0)
if os.environ.get("GTX_X_LIB"): gtx.LIB_PATH = os.environ["GTX_X_LIB"]     # (kernel experiments: a variant build)
eng = gtx.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.profile(True)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
reads = reads[torch.randperm(n, device=dev)]
eng.set_refs(synth.genome_intervals(1_000_000, 43, 50, 2000), 24)
hits = torch.zeros(eng.n_refs, dtype=torch.int64, device=dev)
w = torch.randint(0, 5, (n,), dtype=torch.int32, device=dev) if weighted else None
for _ in range(5): eng.count_device(reads.data_ptr(), n, hits.data_ptr(), w.data_ptr() if weighted else None, 0)
eng.sync(); print("bucket path%s: %.3f ms" % (" (label weights)" if weighted else "", np.mean([eng.profile_last(b)[0] for b in range(3)])))
