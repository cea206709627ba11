#!/usr/bin/env python3
"""Randomised stress of the count / coverage / scan paths at sizes between the unit tests and the bench
(10^5..3x10^6 reads, up to 3x10^5 regions, 1..400 classes, several orders and shapes) against the CPU oracle."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np, gtx
from oracle import orc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
e = gtx.Engine(0)
t0 = time.time()
for it in range(rounds):
    rng = np.random.default_rng(seed0 * 1000 + it)
    n = int(rng.choice([100_000, 400_000, 1_000_000, 3_000_000]))
    m = int(rng.choice([1, 50, 5_000, 60_000, 300_000]))
    ncls = int(rng.choice([1, 3, 24, 48, 400]))
    span = int(rng.choice([2_000, 1_000_000, 200_000_000]))
    rlen = int(rng.choice([1, 36, 150, 5000]))
    shape = rng.choice(["uniform", "clustered", "few-classes"])
    rc = rng.integers(0, ncls, size=m)
    rs = rng.integers(1, span, size=m)
    rl = rng.choice([0, 1, 50, 2000], size=m) + rng.integers(0, 50, size=m)
    n_long = min(m, 20)                                            # a few very long regions; more would make the oracle quadratic
    rl[rng.integers(0, m, size=n_long)] = span // 3
    refs = np.stack([rc, rs, np.minimum(rs + rl, 2**31 - 10)], axis=1).astype(np.int32)
    qc = rng.integers(0, ncls if shape != "few-classes" else max(1, ncls // 8), size=n)
    if shape == "clustered":
        centers = rng.integers(1, span, size=20)
        qs = np.clip(centers[rng.integers(0, 20, size=n)] + rng.integers(-300, 300, size=n), 1, span)
    else:
        qs = rng.integers(1, span, size=n)
    ql = rlen if rng.random() < 0.5 else rng.integers(0, rlen + 1, size=n)
    reads = np.stack([qc, qs, np.minimum(qs + ql, 2**31 - 10)], axis=1).astype(np.int32)
    order = rng.choice(["sorted", "by-pos", "shuffled", "sorted-blocks"])
    if order == "sorted":
        reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    elif order == "by-pos":
        reads = reads[np.argsort(reads[:, 1], kind="stable")]
    elif order == "sorted-blocks":
        reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
        blocks = np.array_split(np.arange(n), 37)
        reads = reads[np.concatenate([blocks[i] for i in rng.permutation(37)])]
    w = rng.integers(0, 5, size=n).astype(np.int32) if rng.random() < 0.3 else None
    # expected (read, region) pairs the bin-index oracle has to visit: keep it to seconds
    dens = (rl.astype(np.float64) + rlen).sum() / (float(span) * ncls)
    if dens * n > 3e8:
        print("round %d skipped (oracle cost)" % it, flush=True)
        continue
    e.set_refs(refs, ncls)
    want = orc.count(refs, reads, w, algo=orc.BIN_INDEX)
    for flags in (gtx.READS_SORTED, 0):
        got, _ = e.count(reads, w, flags)
        assert np.array_equal(got, want), ("count", it, n, m, ncls, span, rlen, shape, order, flags)
    cov, _ = e.coverage(reads, w)
    assert np.array_equal(cov, orc.coverage(refs, reads, w, algo=orc.BIN_INDEX)), ("coverage", it, n, m, ncls, span, order)
    lens = np.full(ncls, span + 100, dtype=np.int32)
    step = int(rng.choice([25, 1000])); size = step * int(rng.choice([1, 20]))
    if (span + 100) // step * ncls < 60_000_000:
        win, _ = e.scan(reads, lens, step, size, "1", w)
        ww, _ = orc.scan(reads, lens, step, size, "1", w, algo=0)
        assert np.array_equal(win, ww), ("scan", it, n, ncls, span, step, size)
    print("round %d ok: n=%d m=%d classes=%d span=%d len=%d %s %s weights=%s (%.0fs)" % (it, n, m, ncls, span, rlen, shape, order, w is not None, time.time() - t0), flush=True)
print("stress ok")
