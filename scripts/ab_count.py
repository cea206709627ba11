#!/usr/bin/env python3
"""Same-box A/B of the streaming count kernel: one process per configuration (library + environment), alternating, several rounds.
usage: ab_count.py [--rounds N] [--reads N] [--refs M] name=LIB[,ENV=VAL...] ...   (LIB '-' = the tree's libgtx.so)
Prints per configuration the kernel time (events, median of the medians) and the step time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "ibm-cbc-genomic-tools_amd"))
import numpy as np, torch, bench, gtx
from gtx import synth
n, m = %(reads)d, %(refs)d
dev = torch.device("cuda", 0)
refs = synth.genome_intervals(m, 43, 50, 2000)
reads = bench.make_reads_on_device(0, np.arange(24), 1000, dev, per=synth.apportion(n, synth.CHROM_LEN))
hits = torch.zeros(len(refs), dtype=torch.int64, device=dev)
eng = gtx.Engine(0); eng.set_refs(refs, synth.n_classes()); eng.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(5): eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
eng.profile(True)
for _ in range(20): eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
eng.sync()
k = sorted(eng.profile_last(b)[0] for b in range(20))
eng.profile(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("RESULT %%.5f %%.5f %%.5f %%d" %% (k[10], k[2], dt * 1e3, int(hits.sum().item())))
'''


def main():
    args = sys.argv[1:]
    rounds, reads, refs = 3, 100_000_000, 1_000_000
    cfgs = []
    while args:
        a = args.pop(0)
        if a == "--rounds": rounds = int(args.pop(0))
        elif a == "--reads": reads = int(args.pop(0))
        elif a == "--refs": refs = int(args.pop(0))
        else:
            name, rest = a.split("=", 1)
            parts = rest.split(",")
            env = dict(p.split("=", 1) for p in parts[1:])
            cfgs.append((name, parts[0], env))
    res = {c[0]: [] for c in cfgs}
    code = WORKER % {"root": ROOT, "reads": reads, "refs": refs}
    for r in range(rounds):
        for name, lib, env in cfgs:
            e = dict(os.environ); e.update(env)
            if lib != "-":
                e["GTX_LIB_PATH"] = os.path.join(ROOT, lib)
            out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(name, "FAILED", out.stderr[-400:]); continue
            f = line[0].split()
            res[name].append((float(f[1]), float(f[2]), float(f[3]), int(f[4])))
            print(name, f[1:], flush=True)
    print("---- kernel ms (median per run -> median over runs) | p10 | step ms | checksum")
    for name, v in res.items():
        if v:
            v.sort()
            print("%-28s %.4f  %.4f  %.4f  %d" % (name, v[len(v) // 2][0], min(x[1] for x in v), sorted(x[2] for x in v)[len(v) // 2], v[0][3]))


if __name__ == "__main__":
    main()
