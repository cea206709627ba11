#!/usr/bin/env python3
"""count without -gaps when part of the 1 M index regions are multi-interval: 100 M sorted plain reads resident in HBM
(gtx_count_device), with and without gtx_set_ref_blocks; and 10 M spliced reads through gtx_count_add_regions."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
from gtx import synth
from bench import make_reads_on_device
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100_000_000
frac = float(os.environ.get("MULTI_FRAC", "0.3"))
dev = torch.device("cuda", 0)
eng = gtx.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
reads = make_reads_on_device(n, np.arange(24), 1000, dev)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
rng = np.random.default_rng(5)
multi = rng.random(len(refs)) < frac
# a multi-interval region: its envelope cut into 2..8 exons with introns between them
first = np.zeros(len(refs) + 1, dtype=np.int64); blocks = []
nb = np.where(multi, rng.integers(2, 9, size=len(refs)), 1)
for k in range(len(refs)):
    s, e = int(refs[k, 1]), int(refs[k, 2])
    if nb[k] == 1 or e - s < 4 * nb[k]:
        blocks.append((s, e))
    else:
        cuts = np.sort(rng.choice(np.arange(s + 1, e), size=2 * nb[k] - 2, replace=False))
        pts = [s] + cuts.tolist() + [e]
        for j in range(nb[k]): blocks.append((pts[2 * j], pts[2 * j + 1] - (1 if j < nb[k] - 1 else 0)))
    first[k + 1] = len(blocks)
blocks = np.array(blocks, dtype=np.int32)
eng.set_refs(refs, 24)
hits = torch.zeros(eng.n_refs, dtype=torch.int64, device=dev)
def timed(label):
    for _ in range(3): eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    print("%s: %.3f ms per call, checksum %d" % (label, ms, int(hits.sum().item())), flush=True)
    return ms
timed("envelopes only (-gaps rule)")
t0 = time.perf_counter(); eng.set_ref_blocks(first, blocks); print("gtx_set_ref_blocks: %.1f ms (%d multi-interval regions, %d intervals)" % ((time.perf_counter() - t0) * 1e3, int((np.diff(first) > 1).sum()), len(blocks)))
timed("with interval lists (%d %% of the regions multi-interval)" % int(frac * 100))
# spliced reads: 10 M queries of 2-3 blocks
m = min(n // 10, 10_000_000)
q = synth.genome_intervals(m, 91, 200, 3000)
qb = np.zeros((m, 2, 2), dtype=np.int32)
qb[:, 0, 0] = q[:, 1]; qb[:, 0, 1] = q[:, 1] + 30; qb[:, 1, 0] = q[:, 2] - 30; qb[:, 1, 1] = q[:, 2]
qfirst = np.arange(m + 1, dtype=np.int64) * 2
t0 = time.perf_counter()
h, _ = eng.count_stream([], regions=[(q, None, qfirst, qb.reshape(-1, 2))])
print("gtx_count_add_regions: %d spliced reads in %.1f ms (host lists + copy + kernel + result), %d hits" % (m, (time.perf_counter() - t0) * 1e3, int(h.sum())))
