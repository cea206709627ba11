import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(R,'ibm-cbc-genomic-tools_amd')); sys.path.insert(0,R)
import numpy as np, torch, gtx
from gtx import synth
e = gtx.Engine(0)
t=time.time()
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
e.set_refs(refs, 24); print("set_refs", time.time()-t, flush=True)
N=100_000_000
g = torch.Generator(device='cuda'); g.manual_seed(1)
per = synth.apportion(N, synth.CHROM_LEN)
parts=[]
for ci,cnt in enumerate(per):
    s = torch.randint(1, int(synth.CHROM_LEN[ci])-60, (int(cnt),), device='cuda', generator=g, dtype=torch.int32)
    s,_ = torch.sort(s)
    parts.append(torch.stack([torch.full_like(s, ci), s, s+49], dim=1))
reads = torch.cat(parts).contiguous(); del parts
print("reads", reads.shape, flush=True)
hits = torch.zeros(len(refs), dtype=torch.int64, device='cuda')
e.set_stream(torch.cuda.current_stream().cuda_stream)
e.profile(True)
for it in range(8):
    e.count_device(reads.data_ptr(), N, hits.data_ptr()); e.sync()
    k, tot = e.profile_last()
    print("iter", it, "kernel ms", k, "total ms", tot, "GB/s", 12*N/k/1e6, flush=True)
print(e.last_info(), int(hits.sum()))
# unsorted-kernel timing
e.count_device(reads.data_ptr(), N, hits.data_ptr(), flags=0); e.sync(); print("search kernel", e.profile_last())
