"""The C ABI's error contract on a live device: status codes + gtx_last_error text for bad arguments, wrong call
order and out-of-range data; a context stays usable after an error; reference sets can be replaced at will."""
import ctypes

import numpy as np
import pytest

import gtx
from gtx import perm, synth
from oracle import orc, porc

pytestmark = pytest.mark.gpu

E_ARG, E_HIP, E_STATE, E_RANGE = -1, -2, -3, -4


def last(lib, ctx):
    return lib.gtx_last_error(ctx).decode()


def test_call_order_and_arguments():
    lib = gtx.load()
    ctx = lib.gtx_create(0)
    assert ctx
    reads = np.array([[0, 1, 10]], dtype=np.int32)
    out = np.zeros(4, dtype=np.uint64)
    info = gtx.CountInfo()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.gtx_count(ctx, p(reads), None, 1, 0, p(out), ctypes.byref(info)) == E_STATE and "gtx_set_refs has not been called" in last(lib, ctx)
    assert lib.gtx_count_begin(ctx) == E_STATE
    assert lib.gtx_count_add(ctx, p(reads), None, 1, 0) == E_STATE and "gtx_count_begin" in last(lib, ctx)
    refs = np.array([[0, 5, 20], [1, 1, 3]], dtype=np.int32)
    assert lib.gtx_set_refs(ctx, p(refs), 2, 2) == 0
    assert lib.gtx_count(ctx, None, None, 1, 0, p(out), ctypes.byref(info)) == E_ARG
    assert lib.gtx_count(ctx, p(reads), None, -1, 0, p(out), ctypes.byref(info)) == E_ARG
    assert lib.gtx_count(ctx, p(reads), None, 1, 0, None, ctypes.byref(info)) == E_ARG
    assert lib.gtx_count_end(ctx, p(out), ctypes.byref(info)) == E_STATE
    # ... and the context still works
    assert lib.gtx_count(ctx, p(reads), None, 1, gtx.READS_SORTED, p(out), ctypes.byref(info)) == 0
    assert out[:2].tolist() == [1, 0]
    lens = np.array([100, 100], dtype=np.int32); off = np.zeros(2, dtype=np.int64)
    assert lib.gtx_scan(ctx, p(reads), None, 1, p(lens), 2, 10, 25, b"1", 0, p(out), p(off)) == E_ARG and "multiple of window step" in last(lib, ctx)
    assert lib.gtx_scan(ctx, p(reads), None, 1, p(lens), 2, 10, 20, b"p", 0, p(out), p(off)) == E_ARG and "preprocess" in last(lib, ctx)
    assert lib.gtx_scan(ctx, p(reads), None, 1, None, 0, 10, 20, b"1", 0, p(out), p(off)) == E_ARG
    lib.gtx_destroy(ctx)


def test_reference_data_out_of_range():
    lib = gtx.load()
    ctx = lib.gtx_create(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for bad, msg in (([[0, 5, 2**31 - 2]], "coordinate"), ([[-2, 5, 10]], "negative class"), ([[3, 5, 10]], "class id >= n_classes")):
        refs = np.array(bad, dtype=np.int32)
        assert lib.gtx_set_refs(ctx, p(refs), 1, 2) == E_RANGE and msg in last(lib, ctx)
    # after a rejected set, counting must refuse or work -- not crash:
    out = np.zeros(2, dtype=np.uint64); info = gtx.CountInfo()
    reads = np.array([[0, 1, 10]], dtype=np.int32)
    rc = lib.gtx_count(ctx, p(reads), None, 1, 0, p(out), ctypes.byref(info))
    assert rc in (0, E_STATE)
    lib.gtx_destroy(ctx)


def test_reference_sets_can_be_replaced(engine):
    rng = np.random.default_rng(3)
    for m, ncls in ((10, 1), (50_000, 24), (3, 2), (0, 1), (200_000, 48), (1, 1)):
        refs = synth.genome_intervals(m, 90 + m % 7, 50, 3000, stranded=ncls == 48) if ncls >= 24 else \
            np.stack([rng.integers(0, ncls, size=m), np.sort(rng.integers(1, 5000, size=m)), np.zeros(m)], axis=1).astype(np.int32)
        if ncls < 24 and m:
            refs[:, 2] = refs[:, 1] + rng.integers(0, 300, size=m)
        reads = synth.genome_intervals(30_000, 91, 40, 41, stranded=ncls == 48) if ncls >= 24 else \
            np.stack([rng.integers(0, ncls, size=2000), rng.integers(1, 5000, size=2000), np.zeros(2000)], axis=1).astype(np.int32)
        if ncls < 24:
            reads[:, 2] = reads[:, 1] + 30
        engine.set_refs(refs, ncls)
        for flags in (0, gtx.READS_SORTED):
            got, _ = engine.count(reads, None, flags)
            np.testing.assert_array_equal(got, orc.count(refs, reads, algo=orc.BIN_INDEX))
        cov, _ = engine.coverage(reads)
        np.testing.assert_array_equal(cov, orc.coverage(refs, reads, algo=orc.BIN_INDEX))


def test_permutation_abi_errors():
    lib = perm._lib()
    h = ctypes.c_void_p()
    assert lib.gtx_perm_create(0, ctypes.byref(h)) == 0
    Y = np.zeros(4); cnt = np.zeros(4, dtype=np.uint64)
    assert lib.gtx_perm_statistic(h, 0, 0, Y.ctypes.data) == E_STATE and "gtx_perm_set_table" in lib.gtx_perm_last_error(h).decode()
    col_ptr = np.array([0, 2, 3], dtype=np.int64); rows = np.array([0, 5, 1], dtype=np.int32)
    V = np.ones(4, dtype=np.float32); sums = np.array([4.0, 4.0, 4.0, 4.0])
    assert lib.gtx_perm_set_table(h, 4, 2, col_ptr.ctypes.data, rows.ctypes.data, V.ctypes.data, None, sums.ctypes.data, 1) == E_RANGE
    rows[1] = 3
    assert lib.gtx_perm_set_table(h, 4, 2, col_ptr.ctypes.data, rows.ctypes.data, V.ctypes.data, None, sums.ctypes.data, 0) == 0
    assert lib.gtx_perm_statistic(h, 6, 0, Y.ctypes.data) == E_ARG and "corr" in lib.gtx_perm_last_error(h).decode()
    assert lib.gtx_perm_statistic(h, 9, 0, Y.ctypes.data) == E_ARG
    assert lib.gtx_perm_count_ge(h, 0, 0, Y.ctypes.data, 1, -1, 10, cnt.ctypes.data) == E_ARG
    tab_ptr = np.array([0, 2, 3], dtype=np.int64); tab = np.ones(3)                 # category 0 has 2 rows: needs 3 entries
    assert lib.gtx_perm_count_rank(h, 0, tab_ptr.ctypes.data, tab.ctypes.data, Y.ctypes.data, 1, 0, 10, cnt.ctypes.data) == E_ARG
    assert lib.gtx_perm_statistic(h, 0, 0, Y.ctypes.data) == 0 and Y[:2].tolist() == [1.0, 1.0]
    lib.gtx_perm_destroy(h)


def test_permutation_tables_can_be_replaced():
    e = perm.PermEngine(0)
    for n_rows, n_cols, seed in ((50, 5, 1), (20000, 300, 2), (7, 2, 3), (3000, 40, 4)):
        t = perm.PermTable.synthetic(n_rows, n_cols, max(2, n_rows // 20), seed=seed, values="gamma", totals=seed % 2 == 0)
        e.set_table(t)
        Y = e.statistic("sum")
        assert np.array_equal(Y.view(np.uint64), porc.statistic(t, "sum").view(np.uint64))
        np.testing.assert_array_equal(e.count_ge("sum", Y, 5, 0, 130), porc.count_ge(t, "sum", Y, 5, 0, 130))
    e.close()


def test_profile_modes(engine):
    """gtx_profile_enable: 1 = events around every device call (kernel + whole call), N >= 2 = kernel-only events on
    every N-th call; the counts are the same with and without the instrumentation."""
    import torch
    refs = synth.genome_intervals(20_000, 5, 50, 2000)
    reads = synth.genome_intervals(300_000, 6, 50, 51)
    engine.set_refs(refs, 24)
    want = orc.count(refs, reads, algo=orc.SORTED_MERGE)
    d_reads = torch.from_numpy(reads).cuda()
    d_hits = torch.zeros(len(refs), dtype=torch.int64, device="cuda")
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for mode, calls, expect in ((1, 5, 5), (4, 9, 3), (0, 2, 0)):
            engine.profile(mode)
            for _ in range(calls):
                engine.count_device(d_reads.data_ptr(), len(reads), d_hits.data_ptr())
            engine.sync()
            np.testing.assert_array_equal(d_hits.cpu().numpy().view(np.uint64), want)
            assert engine.profiled_calls() == expect
            for b in range(expect):
                k, tot = engine.profile_last(b)
                assert 0 < k <= tot if mode == 1 else (k > 0 and tot == k)
            with pytest.raises(gtx.GtxError):
                engine.profile_last(expect)
    finally:
        engine.profile(False)
        engine.set_stream(0)
