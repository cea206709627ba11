#!/usr/bin/env python3
"""Regions into position order: gtx_sort_device on 100 M shuffled reads resident in HBM (the whole call, by the host's clock: the call
returns when the result is complete), and the sortbed tool against sort(1) on the same BED text (N lines, default 20 M)."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, torch, gtx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda"); g.manual_seed(5)
cls = torch.randint(0, 24, (n,), device=dev, dtype=torch.int32, generator=g)
start = torch.randint(1, 240_000_000, (n,), device=dev, dtype=torch.int32, generator=g)
tri = torch.stack([cls, start, start + 49], dim=1).contiguous()
order = torch.empty(n, device=dev, dtype=torch.int32); out = torch.empty_like(tri)
eng = gtx.Engine(0)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.sort_device(tri.data_ptr(), n, 24, order.data_ptr(), out.data_ptr())
    dt = time.perf_counter() - t0
    print("gtx_sort_device: %d reads in %.2f ms = %.3g reads/s" % (n, dt * 1e3, n / dt), flush=True)
key = out[:, 0].to(torch.int64) * (1 << 32) + out[:, 1].to(torch.int64)
print("sorted:", bool((key[1:] >= key[:-1]).all()))
del key, tri, out, order, cls, start
torch.cuda.empty_cache()
# the tool: N lines of shuffled BED6
BIN = os.path.join(R, "ibm-cbc-genomic-tools_amd", "csrc")
d = os.environ.get("TMPDIR", "/tmp")
src = os.path.join(d, "sortbench_sorted.bed"); shuf = os.path.join(d, "sortbench.bed")
subprocess.run([os.path.join(BIN, "gtx_packtool"), "synth", str(lines), "9", src], check=True)
rng = np.random.default_rng(1)
rows = open(src, "rb").read().split(b"\n")[:-1]
perm = rng.permutation(len(rows))
with open(shuf, "wb") as f:
    f.write(b"\n".join(rows[i] for i in perm) + b"\n")
del rows
for name, cmd, env in (("sortbed -i", [os.path.join(BIN, "sortbed"), "-i", shuf], None),
                       ("sortbed -i -o x.gtx", [os.path.join(BIN, "sortbed"), "-i", "-o", os.path.join(d, "sortbench.gtx"), shuf], None),
                       ("LC_ALL=C sort -k1,1 -k2,2n", ["sort", "-k1,1", "-k2,2n", shuf], dict(os.environ, LC_ALL="C")),
                       ("LC_ALL=C sort --parallel=16 -S 8G", ["sort", "--parallel=16", "-S", "8G", "-k1,1", "-k2,2n", shuf], dict(os.environ, LC_ALL="C"))):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, stdout=open(os.path.join(d, "sortbench.out"), "wb"), stderr=subprocess.PIPE, env=env)
    dt = time.perf_counter() - t0
    import hashlib
    h = hashlib.md5(open(os.path.join(d, "sortbench.out"), "rb").read()).hexdigest()[:12]
    print("%-36s %d lines in %.2f s (rc %d, md5 of stdout %s)" % (name, lines, dt, r.returncode, h), flush=True)
