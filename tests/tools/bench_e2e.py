#!/usr/bin/env python3
"""End-to-end timings beside the kernel-only number ("timed three ways"):
  (ii)  packed host arrays -> counts on the host (gtx_count: H2D + kernels + D2H)
  (iii) BED text -> stdout: the product CLI (C++ packer threads + GPU) vs the CPU oracle CLI
        (single thread, the restated reference algorithms incl. text parsing)
Usage: bench_e2e.py [n_reads] [n_refs]"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, pandas as pd
import gtx
from gtx import synth
from oracle import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
tmp = os.environ.get("TMPDIR", "/tmp")
refs = synth.genome_intervals(m, 43, 50, 2000)
reads = synth.genome_intervals(n, 44, 50, 51)
print("cores available: %d" % len(os.sched_getaffinity(0)), flush=True)

eng = gtx.Engine(0)
eng.set_refs(refs, 24)
eng.count(reads[:1000])
t = time.perf_counter(); hits, _ = eng.count(reads); dt = time.perf_counter() - t
print("(ii) packed host -> counts: %.3f s  = %.3g reads/s (12 B/read over PCIe: %.1f GB/s)" % (dt, n / dt, 12 * n / dt / 1e9), flush=True)

names = np.array(synth.CHROM_NAMES)
def write(path, tri, label):
    df = pd.DataFrame({"c": names[tri[:, 0]], "s": tri[:, 1] - 1, "e": tri[:, 2]})
    if label: df["l"] = ["g%d" % i for i in range(len(tri))]
    df.to_csv(path, sep="\t", header=False, index=False)
rp, qp = os.path.join(tmp, "e2e_refs.bed"), os.path.join(tmp, "e2e_reads.bed")
write(rp, refs, True); write(qp, reads, False)
print("BED text: reads %.2f GB, refs %.1f MB" % (os.path.getsize(qp) / 1e9, os.path.getsize(rp) / 1e6), flush=True)
exe = os.path.join(R, "ibm-cbc-genomic-tools_amd", "csrc", "genomic_overlaps")
outs = {}
for name, cmd in (("product CLI  count -S -i", [exe, "count", "-S", "-i", rp, qp]),
                  ("product CLI  count -i   ", [exe, "count", "-i", rp, qp]),
                  ("oracle  CLI  count -S -i", [orc.CLI, "count", "-S", "-i", rp, qp]),
                  ("oracle  CLI  count -i   ", [orc.CLI, "count", "-i", rp, qp])):
    t = time.perf_counter(); r = subprocess.run(cmd, capture_output=True); dt = time.perf_counter() - t
    outs[name] = r.stdout
    print("(iii) %s: %.2f s = %.3g reads/s (rc %d)" % (name, dt, n / dt, r.returncode), flush=True)
vals = list(outs.values())
print("outputs identical:", all(v == vals[0] for v in vals))
os.remove(rp); os.remove(qp)
