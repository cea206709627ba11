#!/usr/bin/env python3
"""permutation_test inner loops on one MI355X: kernel times of gtx_perm_count_ge for a GO-like table."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ibm-cbc-genomic-tools_amd")); sys.path.insert(0, R)
import numpy as np
from gtx import perm
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
mean = int(sys.argv[3]) if len(sys.argv) > 3 else 200
P = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
stats = sys.argv[5].split(",") if len(sys.argv) > 5 else ["sum", "n", "t"]
t0 = time.time()
t = perm.PermTable.synthetic(n_rows, n_cols, mean, seed=1, values="gamma")
nnz = int(t.col_ptr[-1])
print("table %d rows x %d categories, %d memberships (%.1fs to build)" % (n_rows, n_cols, nnz, time.time() - t0), flush=True)
e = perm.PermEngine(0); e.set_table(t)
for stat in stats:
    Y = e.statistic(stat)
    e.count_ge(stat, Y, 1, 0, 256)
    w = time.time(); c = e.count_ge(stat, Y, 1, 0, P); wall = time.time() - w
    a, s = e.last_ms()
    print("%-5s P=%d: apply %.3f ms, stat %.3f ms (%.2f TB/s of 4 B gathers), wall %.1f ms; %.3g member-sums/s" %
          (stat, P, a, s, nnz * P * 4 / (s * 1e-3) / 1e12, wall * 1e3, nnz * P / ((a + s) * 1e-3)), flush=True)
