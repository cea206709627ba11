#!/usr/bin/env python3
"""(With the finalize step on the exchange stream -- the default for reads in stream order -- "kernel+finalize" below is the streaming
kernel alone: the events sit on the member's own stream; "per call back to back" is the number that counts.)
What ONE member of an 8-member group does per call on the fixed 100 M x 1 M shape, timed on one GPU (GTX_GROUP_NO_EXCHANGE=1:
the member exists without a communicator, nothing travels): the member with the largest LPT share, and member 0 (which also puts
the compact vector into file order).  Prints kernel ms / whole-call ms by events; run under rocprofv3 for the per-kernel CSV."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ibm-cbc-genomic-tools_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import gtx  # noqa: E402
from gtx import synth  # noqa: E402

os.environ["GTX_GROUP_NO_EXCHANGE"] = "1"
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
total = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
dev = torch.device("cuda", 0)
refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
per = synth.apportion(total, synth.CHROM_LEN)
owner = gtx.lpt_assign(per, world)
share = [int(per[owner == m].sum()) for m in range(world)]
big = int(np.argmax(share))
order = sorted({big, 0}, reverse=len(sys.argv) > 3 and sys.argv[3] == "rev")
for member in order:
    g = gtx.Group(rank=member, world=world, device=0, unique_id=None)
    g.assign(per)
    g.set_refs(refs, synth.n_classes())
    g.set_stream(member, torch.cuda.current_stream().cuda_stream)
    ch = np.nonzero(owner == member)[0].astype(np.int64)
    reads = bench.make_reads_on_device(0, ch, 1000, dev, per=per[ch])
    hits = [torch.zeros(len(refs), dtype=torch.int64, device=dev) for _ in range(2)]
    for i in range(5):
        g.count_device([reads.data_ptr()], [reads.shape[0]], hits[i & 1].data_ptr())
    g.sync()
    g.profile(member, True)
    for i in range(20):
        g.count_device([reads.data_ptr()], [reads.shape[0]], hits[i & 1].data_ptr())
    g.sync()
    k = sorted(g.profile_last(member, b) for b in range(20))
    g.profile(member, False)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(50):
        g.count_device([reads.data_ptr()], [reads.shape[0]], hits[i & 1].data_ptr())
    th = (time.perf_counter() - t0) / 50                         # what the host spends enqueueing one call
    g.sync(); dt = (time.perf_counter() - t0) / 50
    print("member %d of %d: %d reads (largest share: member %d), kernel %.4f ms, kernel+finalize %.4f ms (medians of 20, events), %.4f ms per call back to back (host enqueue %.4f ms per call)"
          % (member, world, reads.shape[0], big, k[10][0], k[10][1], dt * 1e3, th * 1e3), flush=True)
    g.close()
