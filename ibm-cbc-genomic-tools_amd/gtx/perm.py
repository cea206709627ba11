"""ctypes binding of the permutation-test entry points of libgtx.so (include/gtx_perm.h).

Like the rest of this package: a view of the C ABI for the parity tests and bench.py, no compute
of its own, no CPU path.
"""
import ctypes

import numpy as np

from . import GtxError, load

STAT = {"sum": 0, "n": 1, "sens": 2, "spec": 3, "ratio": 4, "t": 5, "corr": 6}
USE_TOTALS = 1

_vp, _i64, _u64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int
# name -> (restype, argtypes); must list every symbol include/gtx_perm.h declares
ABI = {
    "gtx_perm_create": (_int, [_int, ctypes.POINTER(_vp)]),
    "gtx_perm_destroy": (None, [_vp]),
    "gtx_perm_last_error": (ctypes.c_char_p, [_vp]),
    "gtx_perm_set_table": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint32]),
    "gtx_perm_statistic": (_int, [_vp, _int, _int, _vp]),
    "gtx_perm_count_ge": (_int, [_vp, _int, _int, _vp, _u64, _i64, _i64, _vp]),
    "gtx_perm_count_rank": (_int, [_vp, _int, _vp, _vp, _vp, _u64, _i64, _i64, _vp]),
    "gtx_perm_statistic_approx": (_int, [_vp, _int, _int, _vp]),
    "gtx_perm_count_rank_approx": (_int, [_vp, _int, _int, _vp, _u64, _i64, _i64, _vp]),
    "gtx_perm_permutation": (_int, [_vp, _u64, _i64, _vp]),
    "gtx_perm_last_ms": (_int, [_vp, _vp, _vp]),
}
_typed = False


def _lib():
    global _typed
    lib = load()
    if not _typed:
        for name, (res, args) in ABI.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _typed = True
    return lib


def table_sums(V, Vtotal, use_totals):
    """Vsum, VsumZ, Vsum2, Vtotal_sum as the reference's constructor accumulates them
    (permutation_test.cpp:201-204): sequential double sums over the rows."""
    V = np.asarray(V, dtype=np.float32)
    Vt = np.asarray(Vtotal, dtype=np.float32)

    def seq(x):
        x = np.asarray(x, dtype=np.float64)
        return float(np.cumsum(x)[-1]) if len(x) else 0.0
    if not use_totals:
        return np.array([seq(V), 0.0, seq(V * V), seq(Vt)])           # float32 product, as there
    z = (V / Vt).astype(np.float32)
    zd = z.astype(np.float64)
    return np.array([seq(V), seq(z), seq(zd * zd), seq(Vt)])          # pow(x, 2.0) of a float is exact


class PermTable:
    """Categories as CSR over the rows + per-row values: what StringSets holds after its constructor."""

    def __init__(self, n_rows, col_ptr, rows, V, Vtotal=None, use_totals=True):
        self.n_rows = int(n_rows)
        self.col_ptr = np.ascontiguousarray(col_ptr, dtype=np.int64)
        self.rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.n_cols = len(self.col_ptr) - 1
        self.V = np.ascontiguousarray(V, dtype=np.float32)
        self.Vtotal = np.ones(self.n_rows, dtype=np.float32) if Vtotal is None else np.ascontiguousarray(Vtotal, dtype=np.float32)
        self.has_totals = Vtotal is not None
        self.use_totals = bool(use_totals)
        self.sums = table_sums(self.V, self.Vtotal, self.use_totals)

    @staticmethod
    def synthetic(n_rows, n_cols, mean_size, seed, values="normal", totals=False, use_totals=True):
        """Random membership lists (sizes ~ geometric around mean_size, ascending rows) and values."""
        rng = np.random.default_rng(seed)
        sizes = np.clip(rng.geometric(1.0 / mean_size, size=n_cols), 1, n_rows)
        lists = []
        for c in range(n_cols):
            if n_rows <= 100_000:
                lists.append(np.sort(rng.choice(n_rows, size=sizes[c], replace=False)))
            else:                                                    # large tables: draw with replacement, keep the distinct rows
                lists.append(np.unique(rng.integers(0, n_rows, size=sizes[c])))
        col_ptr = np.zeros(n_cols + 1, dtype=np.int64)
        np.cumsum([len(x) for x in lists], out=col_ptr[1:])
        rows = np.concatenate(lists).astype(np.int32) if lists else np.zeros(0, dtype=np.int32)
        if values == "normal":
            V = rng.normal(size=n_rows)
        elif values == "binary":
            V = (rng.random(n_rows) < 0.15).astype(np.float64)
        elif values == "signed":
            V = rng.integers(-2, 3, size=n_rows).astype(np.float64)
        else:
            V = rng.gamma(2.0, 3.0, size=n_rows)
        Vt = rng.uniform(0.5, 20.0, size=n_rows) if totals else None
        return PermTable(n_rows, col_ptr, rows, V, Vt, use_totals)


class PermEngine:
    def __init__(self, device=0):
        self._lib = _lib()
        h = _vp()
        rc = self._lib.gtx_perm_create(int(device), ctypes.byref(h))
        if rc != 0 or not h:
            raise GtxError("gtx_perm_create(%d) failed (%d): no usable HIP device; there is no CPU path" % (device, rc))
        self._h = h
        self.table = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gtx_perm_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise GtxError("%s failed (%d): %s" % (what, rc, self._lib.gtx_perm_last_error(self._h).decode()))

    def set_table(self, t):
        self.table = t
        vt = t.Vtotal.ctypes.data if t.has_totals else None
        self._check(self._lib.gtx_perm_set_table(self._h, t.n_rows, t.n_cols, t.col_ptr.ctypes.data, t.rows.ctypes.data, t.V.ctypes.data, vt,
                                                 t.sums.ctypes.data, USE_TOTALS if t.use_totals else 0), "gtx_perm_set_table")

    def statistic(self, stat, under=False):
        Y = np.empty(self.table.n_cols, dtype=np.float64)
        self._check(self._lib.gtx_perm_statistic(self._h, STAT[stat], int(under), Y.ctypes.data), "gtx_perm_statistic")
        return Y

    def count_ge(self, stat, Y, seed, first_perm, n_perm, under=False):
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        counts = np.empty(self.table.n_cols, dtype=np.uint64)
        self._check(self._lib.gtx_perm_count_ge(self._h, STAT[stat], int(under), Y.ctypes.data, int(seed), int(first_perm), int(n_perm),
                                                counts.ctypes.data), "gtx_perm_count_ge")
        return counts

    def count_rank(self, tab_ptr, tab, sorted_y, seed, first_perm, n_perm, under=False):
        tab_ptr = np.ascontiguousarray(tab_ptr, dtype=np.int64)
        tab = np.ascontiguousarray(tab, dtype=np.float64)
        sorted_y = np.ascontiguousarray(sorted_y, dtype=np.float64)
        counts = np.empty(self.table.n_cols, dtype=np.uint64)
        self._check(self._lib.gtx_perm_count_rank(self._h, int(under), tab_ptr.ctypes.data, tab.ctypes.data, sorted_y.ctypes.data, int(seed),
                                                  int(first_perm), int(n_perm), counts.ctypes.data), "gtx_perm_count_rank")
        return counts

    def statistic_approx(self, stat, under=False):
        P = np.empty(self.table.n_cols, dtype=np.float64)
        self._check(self._lib.gtx_perm_statistic_approx(self._h, STAT[stat], int(under), P.ctypes.data), "gtx_perm_statistic_approx")
        return P

    def count_rank_approx(self, stat, sorted_y, seed, first_perm, n_perm, under=False):
        sorted_y = np.ascontiguousarray(sorted_y, dtype=np.float64)
        counts = np.empty(self.table.n_cols, dtype=np.uint64)
        self._check(self._lib.gtx_perm_count_rank_approx(self._h, STAT[stat], int(under), sorted_y.ctypes.data, int(seed), int(first_perm),
                                                         int(n_perm), counts.ctypes.data), "gtx_perm_count_rank_approx")
        return counts

    def permutation(self, seed, q):
        out = np.empty(self.table.n_rows, dtype=np.int32)
        self._check(self._lib.gtx_perm_permutation(self._h, int(seed), int(q), out.ctypes.data), "gtx_perm_permutation")
        return out

    def last_ms(self):
        a, b = ctypes.c_float(), ctypes.c_float()
        self._check(self._lib.gtx_perm_last_ms(self._h, ctypes.byref(a), ctypes.byref(b)), "gtx_perm_last_ms")
        return a.value, b.value
