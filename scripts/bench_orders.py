#!/usr/bin/env python3
"""How the streaming kernel behaves on read orders other than (class, start)-sorted, 100M reads x 1M refs on one GPU."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
eng = gtx.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.profile(True)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
def run(name, r, flags=gtx.READS_SORTED, nrefcls=24, refs=None):
    hits = torch.zeros(eng.n_refs, dtype=torch.int64, device=dev)
    for _ in range(3): eng.count_device(r.data_ptr(), r.shape[0], hits.data_ptr(), None, flags)
    eng.sync(); k = np.mean([eng.profile_last(b)[0] for b in range(2)])
    print("%-55s kernel %.3f ms  %.3g reads/s" % (name, k, r.shape[0] / (k * 1e-3)), flush=True)
eng.set_refs(synth.genome_intervals(1_000_000, 43, 50, 2000), 24)
run("sorted by (class,start), walk kernel", reads)
run("same, search kernel (no sorted hint)", reads, 0)
# variable read lengths: ends not monotone
v = reads.clone(); v[:, 2] = v[:, 1] + torch.randint(20, 3000, (n,), device=dev, dtype=torch.int32)
run("sorted starts, read length 20..3000", v)
# strand-aware classes on a position-sorted stream (classes interleave lane by lane)
s = reads.clone(); s[:, 0] += 24 * torch.randint(0, 2, (n,), device=dev, dtype=torch.int32)
eng.set_refs(synth.genome_intervals(1_000_000, 43, 50, 2000, stranded=True), 48)
run("position-sorted, 2 strand classes interleaved", s)
g = s[torch.argsort(s[:, 0], stable=True)]
run("same reads grouped by strand class (what the packer emits)", g)
p = reads[torch.randperm(n, device=dev)]
eng.set_refs(synth.genome_intervals(1_000_000, 43, 50, 2000), 24)
run("random order, walk kernel (hint is wrong)", p[:20_000_000])
run("random order, bucket path (default for unsorted)", p[:20_000_000], 0)
# the two order-agnostic paths side by side (GTX_BUCKET_MIN_READS is read when the context is created)
os.environ["GTX_BUCKET_MIN_READS"] = str(1 << 40)
eng2 = gtx.Engine(0); eng2.set_stream(torch.cuda.current_stream().cuda_stream); eng2.profile(True)
eng2.set_refs(synth.genome_intervals(1_000_000, 43, 50, 2000), 24)
def run2(name, r):
    hits = torch.zeros(eng2.n_refs, dtype=torch.int64, device=dev)
    for _ in range(3): eng2.count_device(r.data_ptr(), r.shape[0], hits.data_ptr(), None, 0)
    eng2.sync(); k = np.mean([eng2.profile_last(b)[0] for b in range(2)])
    print("%-55s kernel %.3f ms  %.3g reads/s" % (name, k, r.shape[0] / (k * 1e-3)), flush=True)
run2("random order, per-read search kernel only", p[:20_000_000])
del v, s, g
pp = reads[torch.randperm(n, device=dev)]
run("random order, all %d reads, bucket path" % n, pp, 0)
run2("random order, all %d reads, search kernel" % n, pp)
