#!/usr/bin/env python3
"""Randomised stress of the permutation-test path: random tables (shapes, value kinds, totals, -norm), all statistics,
-u, random seeds / permutation ranges -- statistics and exceed-counts must equal the CPU oracle bit for bit."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np
from gtx import perm
from oracle import porc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = perm.PermEngine(0)
t0 = time.time()
for it in range(rounds):
    rng = np.random.default_rng(seed0 * 1000 + it)
    n_rows = int(rng.choice([3, 16, 17, 200, 5000, 40000]))
    n_cols = int(rng.choice([1, 7, 100, 600]))
    mean = max(1, int(rng.choice([2, 20, 200])) if n_rows > 50 else 2)
    kind = str(rng.choice(["normal", "binary", "signed", "gamma"]))
    totals = bool(rng.random() < 0.5)
    use_totals = bool(rng.random() < 0.7)
    t = perm.PermTable.synthetic(n_rows, n_cols, min(mean, n_rows), seed=int(rng.integers(1 << 30)), values=kind, totals=totals, use_totals=use_totals or not totals)
    if not t.use_totals:
        t = perm.PermTable(t.n_rows, t.col_ptr, t.rows, (t.V / t.Vtotal).astype(np.float32), None, use_totals=False)
    e.set_table(t)
    for stat in ("sum", "n", "sens", "spec", "ratio", "t", "corr"):
        if stat == "corr" and not t.use_totals:
            continue
        under = bool(rng.random() < 0.3)
        Y = porc.statistic(t, stat, under)
        got = e.statistic(stat, under)
        assert np.array_equal(np.isnan(got), np.isnan(Y)) and np.array_equal(got[~np.isnan(Y)].view(np.uint64), Y[~np.isnan(Y)].view(np.uint64)), ("stat", it, stat, under)
        seed = int(rng.integers(1 << 62)); first = int(rng.choice([0, 1, 63, 1000])); P = int(rng.choice([1, 64, 130, 400]))
        c = e.count_ge(stat, Y, seed, first, P, under)
        assert np.array_equal(c, porc.count_ge(t, stat, Y, seed, first, P, under)), ("count", it, stat, under, seed, first, P)
    print("round %d ok: rows=%d cols=%d mean=%d %s totals=%s use_totals=%s (%.0fs)" % (it, n_rows, n_cols, mean, kind, totals, t.use_totals, time.time() - t0), flush=True)
print("stress ok")
