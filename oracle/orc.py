"""ctypes wrapper of oracle/libgtx_oracle.so -- TEST INFRASTRUCTURE ONLY (see gtx_oracle.c).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgtx_oracle.so")
CLI = os.path.join(HERE, "gtx_oracle")

BIN_INDEX = 0      # UnsortedGenomicRegionSetOverlaps restatement
SORTED_MERGE = 1   # SortedGenomicRegionSetOverlaps restatement
_lib = None


def build():
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_count_packed.restype = ctypes.c_int
        L.orc_count_packed.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_long, ctypes.c_void_p]
        L.orc_coverage_packed.restype = ctypes.c_int
        L.orc_coverage_packed.argtypes = L.orc_count_packed.argtypes
        L.orc_scan_packed.restype = ctypes.c_int
        L.orc_scan_packed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_char, ctypes.c_int, ctypes.c_long,
                                      ctypes.c_void_p, ctypes.c_void_p]
        L.orc_scan_n_windows.restype = ctypes.c_int64
        L.orc_scan_n_windows.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def count(refs, reads, weights=None, algo=BIN_INDEX, max_label_value=1 << 40):
    refs = np.ascontiguousarray(refs, dtype=np.int32)
    reads = np.ascontiguousarray(reads, dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int32)
    hits = np.zeros(max(len(refs), 1), dtype=np.uint64)
    rc = lib().orc_count_packed(_p(refs), len(refs), _p(reads), _p(w), len(reads), algo, max_label_value, _p(hits))
    if rc:
        raise OracleError(lib().orc_last_error().decode())
    return hits[:len(refs)]


def coverage(refs, reads, weights=None, algo=BIN_INDEX, max_label_value=1 << 40):
    refs = np.ascontiguousarray(refs, dtype=np.int32)
    reads = np.ascontiguousarray(reads, dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int32)
    cov = np.zeros(max(len(refs), 1), dtype=np.uint64)
    rc = lib().orc_coverage_packed(_p(refs), len(refs), _p(reads), _p(w), len(reads), algo, max_label_value, _p(cov))
    if rc:
        raise OracleError(lib().orc_last_error().decode())
    return cov[:len(refs)]


def scan(reads, class_len, win_step, win_size, preprocess="1", weights=None, algo=0, max_label_value=1 << 40):
    reads = np.ascontiguousarray(reads, dtype=np.int32)
    cl = np.ascontiguousarray(class_len, dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int32)
    L = lib()
    off, tot = [], 0
    for ln in cl:
        off.append(tot)
        tot += L.orc_scan_n_windows(int(ln), win_step, win_size)
    off = np.asarray(off, dtype=np.int64)
    out = np.zeros(max(tot, 1), dtype=np.uint64)
    rc = L.orc_scan_packed(_p(reads), _p(w), len(reads), _p(cl), len(cl), win_step, win_size, preprocess.encode()[0:1], algo,
                           max_label_value, _p(out), _p(off))
    if rc:
        raise OracleError(L.orc_last_error().decode())
    return out[:tot], off


def cli(args, stdin=None):
    """Run the oracle CLI; returns (exit code, stdout, stderr)."""
    if not os.path.exists(CLI):
        build()
    r = subprocess.run([CLI] + list(args), input=stdin, capture_output=True)
    return r.returncode, r.stdout.decode(), r.stderr.decode()
