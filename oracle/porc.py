"""ctypes wrapper of oracle/libperm_oracle.so -- TEST INFRASTRUCTURE ONLY (see perm_oracle.c).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libperm_oracle.so")
CLI = os.path.join(HERE, "perm_oracle")
STAT = {"sum": 0, "n": 1, "sens": 2, "spec": 3, "ratio": 4, "t": 5, "corr": 6}
BIJECTION, MT19937 = 0, 1
_lib = None
_vp, _i64, _u64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        L = ctypes.CDLL(LIB)
        L.porc_permutation.argtypes = [_u64, _i64, _i64, _vp]
        L.porc_statistic.argtypes = [_int, _int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
        L.porc_count_ge.argtypes = [_int, _int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _i64, _i64, _vp]
        L.porc_count_rank.argtypes = [_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _i64, _i64, _vp]
        L.porc_statistic_approx.argtypes = [_int, _int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
        L.porc_count_rank_approx.argtypes = [_int, _int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _i64, _i64, _vp]
        L.porc_tdist_Q.restype = ctypes.c_double
        L.porc_tdist_Q.argtypes = [ctypes.c_double, ctypes.c_double]
        L.porc_gauss_Q.restype = ctypes.c_double
        L.porc_gauss_Q.argtypes = [ctypes.c_double]
        L.porc_hypergeom_Q.restype = ctypes.c_double
        L.porc_hypergeom_Q.argtypes = [ctypes.c_long] * 4
        L.porc_hypergeom_table.argtypes = [_int, _i64, _i64, _vp, _vp, _vp, _vp]
        _lib = L
    return _lib


def permutation(seed, p, n):
    out = np.empty(n, dtype=np.int32)
    lib().porc_permutation(int(seed), int(p), int(n), out.ctypes.data)
    return out


def statistic(t, stat, under=False):
    """t: a gtx.perm.PermTable (plain arrays; nothing of the product is called)"""
    Y = np.empty(t.n_cols, dtype=np.float64)
    lib().porc_statistic(STAT[stat], int(under), int(t.use_totals), t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.rows.ctypes.data,
                         t.V.ctypes.data, t.Vtotal.ctypes.data, t.sums.ctypes.data, Y.ctypes.data)
    return Y


def count_ge(t, stat, Y, seed, first_perm, n_perm, under=False, source=BIJECTION):
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    counts = np.empty(t.n_cols, dtype=np.uint64)
    lib().porc_count_ge(STAT[stat], int(under), int(t.use_totals), t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.rows.ctypes.data,
                        t.V.ctypes.data, t.Vtotal.ctypes.data, t.sums.ctypes.data, Y.ctypes.data, int(source), int(seed),
                        int(first_perm), int(n_perm), counts.ctypes.data)
    return counts


def hypergeom_table(t, under=False):
    sizes = np.diff(t.col_ptr)
    tab_ptr = np.zeros(t.n_cols + 1, dtype=np.int64)
    np.cumsum(sizes + 1, out=tab_ptr[1:])
    tab = np.empty(int(tab_ptr[-1]), dtype=np.float64)
    lib().porc_hypergeom_table(int(under), t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.V.ctypes.data, tab_ptr.ctypes.data, tab.ctypes.data)
    return tab_ptr, tab


def count_rank(t, tab_ptr, tab, sorted_y, seed, first_perm, n_perm, under=False, source=BIJECTION):
    sorted_y = np.ascontiguousarray(sorted_y, dtype=np.float64)
    counts = np.empty(t.n_cols, dtype=np.uint64)
    lib().porc_count_rank(int(under), t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.rows.ctypes.data, t.V.ctypes.data,
                          tab_ptr.ctypes.data, tab.ctypes.data, sorted_y.ctypes.data, int(source), int(seed), int(first_perm),
                          int(n_perm), counts.ctypes.data)
    return counts


def statistic_approx(t, stat, under=False):
    """Calc*Statistic(approx = true) of ratio (without totals) / t / corr: the statistic's p-value under its distribution"""
    P = np.empty(t.n_cols, dtype=np.float64)
    lib().porc_statistic_approx(STAT[stat], int(under), int(t.use_totals), t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.rows.ctypes.data,
                                t.V.ctypes.data, t.Vtotal.ctypes.data, t.sums.ctypes.data, P.ctypes.data)
    return P


def count_rank_approx(t, stat, sorted_y, seed, first_perm, n_perm, under=False, source=BIJECTION):
    sorted_y = np.ascontiguousarray(sorted_y, dtype=np.float64)
    counts = np.empty(t.n_cols, dtype=np.uint64)
    lib().porc_count_rank_approx(STAT[stat], int(under), int(t.use_totals), t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.rows.ctypes.data,
                                 t.V.ctypes.data, t.Vtotal.ctypes.data, t.sums.ctypes.data, sorted_y.ctypes.data, int(source), int(seed),
                                 int(first_perm), int(n_perm), counts.ctypes.data)
    return counts


def tdist_Q(t, nu):
    return lib().porc_tdist_Q(float(t), float(nu))


def gauss_Q(x):
    return lib().porc_gauss_Q(float(x))


def hypergeom_Q(k, n1, n2, t):
    return lib().porc_hypergeom_Q(int(k), int(n1), int(n2), int(t))


def run_cli(args, env=None):
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([CLI] + [str(a) for a in args], capture_output=True, env=e)
