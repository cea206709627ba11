"""Parity of the HIP count path (through the C ABI) against the CPU oracle -- bit-exact (uint64)."""
import numpy as np
import pytest

import gtx
from gtx import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def both_oracles(refs, reads, weights=None):
    """The oracle's two restated algorithms (bin index and sorted merge) wherever both apply: reads sorted by
    (class, start) and regions the merge accepts; otherwise the bin index alone."""
    a = orc.count(refs, reads, weights, algo=orc.BIN_INDEX)
    n = len(reads)
    in_order = n < 2 or bool(np.all((reads[1:, 0] > reads[:-1, 0]) | ((reads[1:, 0] == reads[:-1, 0]) & (reads[1:, 1] >= reads[:-1, 1]))))
    refs = np.asarray(refs)
    if in_order and len(refs) and bool(np.all((refs[:, 1] <= refs[:, 2]) & (refs[:, 2] > 0))) and bool(np.all(reads[:, 1] <= reads[:, 2])):
        order = np.lexsort((refs[:, 1], refs[:, 0]))               # the merge wants the index set sorted; counts go back to file order
        b = np.empty_like(a)
        b[order] = orc.count(refs[order], reads, weights, algo=orc.SORTED_MERGE)
        np.testing.assert_array_equal(a, b)
    return a


def check(engine, refs, reads, weights=None, flags=gtx.READS_SORTED, n_classes=0):
    engine.set_refs(refs, n_classes)
    hits, info = engine.count(reads, weights, flags)
    want = both_oracles(refs, reads, weights)
    assert hits.dtype == np.uint64
    np.testing.assert_array_equal(hits, want)
    return hits, info


def test_g2_toy(engine):
    # SURVEY.md 8(c) G2 with -i: A 2 / B 3 / C 1 ; classes chr1=0 chr2=1 chr3=2
    refs = np.array([[0, 101, 200], [0, 151, 300], [1, 11, 20]], dtype=np.int32)
    reads = np.array([[0, 51, 100], [0, 51, 101], [0, 200, 250], [0, 201, 250], [0, 300, 350], [1, 20, 30], [2, 2, 5]], dtype=np.int32)
    for flags in (gtx.READS_SORTED, 0):
        engine.set_refs(refs)
        hits, info = engine.count(reads, None, flags)
        assert hits.tolist() == [2, 3, 1]
        assert info["n_no_class"] == 1          # the chr3 read: unknown chromosome, ignored


def test_g2_toy_strand_aware(engine):
    # strand folded into the class: class = strand*3 + chrom -> A 2 / B 1 / C 1
    refs = np.array([[0, 101, 200], [3, 151, 300], [1, 11, 20]], dtype=np.int32)
    reads = np.array([[0, 51, 100], [0, 51, 101], [0, 200, 250], [0, 201, 250], [1, 20, 30], [2, 2, 5], [3, 300, 350]], dtype=np.int32)
    engine.set_refs(refs, 6)
    hits, _ = engine.count(reads)
    assert hits.tolist() == [2, 1, 1]


@pytest.mark.parametrize("n,m,seed", [(1, 1, 0), (63, 5, 1), (64, 64, 2), (65, 1000, 3), (4097, 3, 4), (100000, 2000, 5), (300000, 50000, 6)])
def test_single_chrom_sorted(engine, n, m, seed):
    refs = synth.refs_single_chrom(m, seed=seed, chrom_len=5_000_000)
    reads = synth.reads_single_chrom(n, seed=seed, chrom_len=5_000_000)
    check(engine, refs, reads)
    check(engine, refs, reads, flags=0)


def test_more_refs_than_reads(engine):
    refs = synth.refs_single_chrom(200000, seed=7, chrom_len=3_000_000)
    reads = synth.reads_single_chrom(5000, seed=7, chrom_len=3_000_000)
    check(engine, refs, reads)


def test_search_kernel_with_coarse_lds_samples(engine):
    """> 2^20 boundaries per array: the LDS top level of the order-agnostic kernel samples every 256th
    boundary instead of every 64th; shuffled reads over 24 chromosomes, one of them without references."""
    rng = np.random.default_rng(77)
    refs = synth.genome_intervals(2_300_000, 5, 50, 2000)
    refs = refs[refs[:, 0] != 3]
    reads = synth.genome_intervals(150_000, 6, 36, 36)
    reads = reads[rng.permutation(len(reads))]
    check(engine, refs, reads, flags=0, n_classes=synth.n_classes())


def test_refs_in_file_order_unsorted_and_overlapping(engine):
    rng = np.random.default_rng(8)
    refs = synth.refs_single_chrom(30000, seed=8, chrom_len=2_000_000, max_len=20000)
    refs = refs[rng.permutation(len(refs))]            # file order != sorted order
    reads = synth.reads_single_chrom(200000, seed=8, chrom_len=2_000_000)
    check(engine, refs, reads)


def test_multi_chrom(engine):
    refs = synth.genome_intervals(40000, 9, 50, 2000)
    reads = synth.genome_intervals(500000, 10, 50, 51)
    check(engine, refs, reads, n_classes=synth.n_classes())
    refs = synth.genome_intervals(40000, 11, 50, 2000, stranded=True)
    reads = synth.genome_intervals(500000, 12, 50, 51, stranded=True)
    check(engine, refs, reads, n_classes=synth.n_classes(True))


def test_variable_length_reads(engine):
    # ends are not monotone in a start-sorted stream: exercises the backward walk of the end window
    rng = np.random.default_rng(13)
    reads = synth.genome_intervals(300000, 13, 20, 30000)
    refs = synth.genome_intervals(30000, 14, 50, 5000)
    check(engine, refs, reads, n_classes=synth.n_classes())
    del rng


def test_unsorted_reads_any_order(engine):
    rng = np.random.default_rng(15)
    refs = synth.genome_intervals(20000, 15, 50, 2000)
    reads = synth.genome_intervals(200000, 16, 50, 51)
    reads = reads[rng.permutation(len(reads))]
    # both kernels must be exact on arbitrary order; the sorted hint is only a speed hint
    check(engine, refs, reads, flags=0, n_classes=synth.n_classes())
    check(engine, refs, reads, flags=gtx.READS_SORTED, n_classes=synth.n_classes())


def test_strand_interleaved_position_sorted(engine):
    # position-sorted stream whose classes alternate (strand-aware run on `sortbed -i` order)
    reads = synth.genome_intervals(100000, 17, 50, 51, stranded=True)
    chrom = reads[:, 0] % 24
    reads = reads[np.lexsort((reads[:, 1], chrom))]
    refs = synth.genome_intervals(10000, 18, 50, 2000, stranded=True)
    check(engine, refs, reads, n_classes=synth.n_classes(True))


def test_weights(engine):
    rng = np.random.default_rng(19)
    refs = synth.genome_intervals(20000, 19, 50, 2000)
    reads = synth.genome_intervals(200000, 20, 50, 51)
    w = rng.integers(0, 4, size=len(reads)).astype(np.int32)
    check(engine, refs, reads, w, n_classes=synth.n_classes())
    check(engine, refs, reads, w, flags=0, n_classes=synth.n_classes())
    # negative label values wrap in unsigned long exactly like the reference's `hits[...] += long`
    w2 = rng.integers(-3, 4, size=len(reads)).astype(np.int32)
    check(engine, refs, reads, w2, n_classes=synth.n_classes())


def test_invalid_refs_are_skipped(engine):
    # start>stop or stop<=0 reference regions stay in the numbering with count 0 (genomic_intervals.cpp:5659)
    refs = np.array([[0, 100, 200], [0, 300, 250], [0, -5, 0], [0, 150, 150], [0, -10, 5]], dtype=np.int32)
    reads = np.array([[0, 1, 1000], [0, 150, 150], [0, 260, 290], [0, 1, 3]], dtype=np.int32)
    hits, _ = check(engine, refs, reads)
    assert hits.tolist() == [2, 0, 0, 2, 2]


def test_degenerate_reads_are_reported_not_counted(engine):
    refs = np.array([[0, 100, 200]], dtype=np.int32)
    reads = np.array([[0, 120, 130], [0, 151, 150], [0, 160, 170]], dtype=np.int32)
    engine.set_refs(refs)
    for flags in (gtx.READS_SORTED, 0):
        hits, info = engine.count(reads, None, flags)
        assert hits.tolist() == [2]
        assert info["n_degenerate"] == 1 and info["first_degenerate"] == 1


def test_empty_inputs(engine):
    refs = synth.refs_single_chrom(100, seed=21, chrom_len=100000)
    engine.set_refs(refs)
    hits, _ = engine.count(np.zeros((0, 3), dtype=np.int32))
    assert hits.tolist() == [0] * 100
    engine.set_refs(np.zeros((0, 3), dtype=np.int32))
    hits, _ = engine.count(synth.reads_single_chrom(1000, seed=21, chrom_len=100000))
    assert len(hits) == 0


def test_hot_reference(engine):
    # skew: most reads pile onto a handful of reference regions
    rng = np.random.default_rng(22)
    refs = synth.refs_single_chrom(5000, seed=22, chrom_len=10_000_000)
    reads = synth.reads_single_chrom(400000, seed=22, chrom_len=10_000_000)
    hot = refs[rng.integers(0, len(refs), size=5)]
    k = len(reads) // 2
    s = hot[rng.integers(0, 5, size=k), 1] + rng.integers(0, 40, size=k)
    reads[:k, 1] = s
    reads[:k, 2] = s + 49
    reads = reads[np.argsort(reads[:, 1], kind="stable")]
    check(engine, refs, reads)


def test_sort_check(engine):
    refs = synth.refs_single_chrom(100, seed=23, chrom_len=1_000_000)
    reads = synth.reads_single_chrom(10000, seed=23, chrom_len=1_000_000)
    engine.set_refs(refs)
    _, info = engine.count(reads, None, gtx.READS_SORTED | gtx.CHECK_SORTED)
    assert info["first_unsorted"] == -1
    bad = reads.copy()
    bad[[7000, 7001]] = bad[[7001, 7000]]
    if bad[7000, 1] == bad[7001, 1]:
        bad[7001, 1] -= 1
    _, info = engine.count(bad, None, gtx.READS_SORTED | gtx.CHECK_SORTED)
    assert info["first_unsorted"] == 7001
    # the streaming fast path checks whole steps of 256 reads: violations inside a step, at a step seam, at a seam between
    # two waves' spans (512 reads each at this size) and in the partial last step, with and without weights
    w = np.ones(len(reads), dtype=np.int32)
    for at in (1, 100, 255, 256, 511, 512, 513, 4095, 4096, 9990, 9999):
        bad = reads.copy()
        bad[[at - 1, at]] = bad[[at, at - 1]]
        if bad[at - 1, 1] == bad[at, 1]:
            bad[at, 1] -= 1
        for ww in (None, w):
            hits, info = engine.count(bad, ww, gtx.READS_SORTED | gtx.CHECK_SORTED)
            assert info["first_unsorted"] == at
            np.testing.assert_array_equal(hits, orc.count(refs, bad, algo=orc.BIN_INDEX))
        hits, info = engine.count(reads, w, gtx.READS_SORTED | gtx.CHECK_SORTED)
        assert info["first_unsorted"] == -1


def test_device_entry_matches_host_entry(engine):
    torch = pytest.importorskip("torch")
    refs = synth.genome_intervals(30000, 24, 50, 2000)
    reads = synth.genome_intervals(400000, 25, 50, 51)
    engine.set_refs(refs, synth.n_classes())
    want, _ = engine.count(reads)
    d_reads = torch.from_numpy(reads).cuda()
    d_hits = torch.zeros(len(refs), dtype=torch.int64, device="cuda")
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    engine.count_device(d_reads.data_ptr(), len(reads), d_hits.data_ptr())
    engine.sync()
    engine.set_stream(0)
    np.testing.assert_array_equal(d_hits.cpu().numpy().view(np.uint64), want)


def test_config2_10m_reads_200k_refs(engine):
    """BASELINE config 2: 10M 50bp reads x 200k exons, one chromosome, bit-exact vs the CPU oracle."""
    refs = synth.refs_single_chrom(200_000, seed=42)
    reads = synth.reads_single_chrom(10_000_000, seed=42)
    engine.set_refs(refs)
    hits, info = engine.count(reads)
    want = orc.count(refs, reads, algo=orc.SORTED_MERGE)
    np.testing.assert_array_equal(hits, want)
    assert info["n_degenerate"] == 0 and info["n_no_class"] == 0


def test_streaming_begin_add_end(engine):
    """gtx_count_begin / _add / _end over several batches == one gtx_count; the order check works across seams."""
    refs = synth.genome_intervals(20000, 61, 50, 2000)
    reads = synth.genome_intervals(300000, 62, 50, 51)
    engine.set_refs(refs, synth.n_classes())
    want, _ = engine.count(reads)
    cuts = [0, 1, 64, 70000, 70001, 199999, len(reads)]
    batches = [(reads[a:b], None) for a, b in zip(cuts[:-1], cuts[1:])]
    got, info = engine.count_stream(batches, gtx.READS_SORTED | gtx.CHECK_SORTED)
    np.testing.assert_array_equal(got, want)
    assert info["first_unsorted"] == -1
    # a violation exactly at a seam: the first read of a batch sorts before the last read of the previous one
    bad = reads.copy()
    bad[70000], bad[69999] = reads[69999].copy(), reads[70000].copy()
    if bad[70000, 1] == bad[69999, 1]:
        bad[70000, 1] -= 1
    batches = [(bad[a:b], None) for a, b in zip(cuts[:-1], cuts[1:])]
    _, info = engine.count_stream(batches, gtx.READS_SORTED | gtx.CHECK_SORTED)
    assert info["first_unsorted"] == 70000


def test_device_batching_inside_one_call():
    """GTX_BATCH_READS forces gtx_count / gtx_scan / gtx_coverage to split one call into many device batches."""
    import os
    os.environ["GTX_BATCH_READS"] = "50000"
    try:
        e = gtx.Engine(0)
    finally:
        del os.environ["GTX_BATCH_READS"]
    rng = np.random.default_rng(63)
    refs = synth.genome_intervals(8000, 63, 50, 3000)
    reads = synth.genome_intervals(260001, 64, 30, 400)
    w = rng.integers(0, 4, size=len(reads)).astype(np.int32)
    e.set_refs(refs, synth.n_classes())
    hits, info = e.count(reads, w, gtx.READS_SORTED | gtx.CHECK_SORTED)
    np.testing.assert_array_equal(hits, orc.count(refs, reads, w))
    assert info["first_unsorted"] == -1
    cov, _ = e.coverage(reads, w)
    np.testing.assert_array_equal(cov, orc.coverage(refs, reads, w))
    win, _ = e.scan(reads, synth.CHROM_LEN, 1000, 3000)
    want, _ = orc.scan(reads, synth.CHROM_LEN, 1000, 3000)
    np.testing.assert_array_equal(win, want)
    e.close()


def test_linearity_at_large_scale(engine):
    """Size-independent property at a size the CPU oracle is not run at: counts are additive over any split
    of the read stream, and the grand total equals the number of (read, region) overlaps summed per part.
    250M reads (3 GB) resident in HBM, 1M regions."""
    torch = pytest.importorskip("torch")
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = 250_000_000
    dev = torch.device("cuda", 0)
    reads = bench.make_reads_on_device(n, np.arange(24), 77, dev)
    refs = synth.genome_intervals(1_000_000, 43, 50, 2000)
    engine.set_refs(refs, 24)
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    whole = torch.zeros(len(refs), dtype=torch.int64, device=dev)
    part = torch.zeros_like(whole)
    acc = torch.zeros_like(whole)
    engine.count_device(reads.data_ptr(), n, whole.data_ptr())
    cuts = [0, 1, 99_999_937, 100_000_000, 250_000_000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        engine.count_device(reads[a:b].data_ptr(), b - a, part.data_ptr())
        acc += part
    engine.sync()
    engine.set_stream(0)
    assert torch.equal(whole, acc)
    assert engine.last_info()["n_degenerate"] == 0
    # spot-check one chromosome's regions against the oracle on that chromosome's reads
    c = 20                                            # chr7 in strcmp order -> a mid-sized class
    sel = reads[:, 0] == c
    sub = reads[sel].cpu().numpy()
    rsel = np.nonzero(refs[:, 0] == c)[0]
    want = orc.count(refs[rsel], sub, algo=orc.SORTED_MERGE)
    np.testing.assert_array_equal(whole.cpu().numpy().view(np.uint64)[rsel], want)


def test_span_starts_among_equal_boundaries(engine):
    """Every wave places its two windows with the paired two-hop search (rank_pair) through the every-256th-boundary arrays:
    long runs of equal boundaries (ties across many samples), a class above 65 k boundaries (strided first hop) and
    classes smaller than one sample distance must all give the ranks of the plain searches."""
    rng = np.random.default_rng(17)
    parts = []
    for cls, m in ((0, 150_000), (1, 90), (2, 300), (3, 20_000)):
        s = np.sort(rng.integers(1, 4_000_000, size=m))
        parts.append(np.stack([np.full(m, cls), s, s + rng.integers(0, 1500, size=m)], axis=1))
    dup = np.stack([np.zeros(70_000), np.full(70_000, 2_000_000), np.full(70_000, 2_000_010)], axis=1)       # 70 k identical regions
    dup2 = np.stack([np.full(40_000, 3), np.full(40_000, 1_000_000), 1_000_000 + rng.integers(0, 3, size=40_000)], axis=1)
    refs = np.concatenate(parts + [dup, dup2]).astype(np.int32)
    refs = refs[rng.permutation(len(refs))]                                  # file order is not sorted order
    n = 400_000
    reads = np.stack([np.sort(rng.integers(0, 4, size=n)), rng.integers(1, 4_000_000, size=n), np.zeros(n, dtype=np.int64)], axis=1)
    order = np.lexsort((reads[:, 1], reads[:, 0]))
    reads = reads[order]
    reads[:, 2] = reads[:, 1] + 49
    reads = reads.astype(np.int32)
    engine.set_refs(refs, 4)
    hits, info = engine.count(reads, None, gtx.READS_SORTED)
    np.testing.assert_array_equal(hits, orc.count(refs, reads, algo=orc.BIN_INDEX))
    cov, _ = engine.coverage(reads)
    np.testing.assert_array_equal(cov, orc.coverage(refs, reads, algo=orc.BIN_INDEX))


def test_weighted_fast_path_and_its_fallbacks(engine):
    """Weighted sorted reads go in steps of 256 with the weights' prefix sums in LDS when the keys are ordered and the
    weights are small; anything else (huge label values, read ends out of order, class changes) takes the general code.
    All mixes must give the oracle's counts; dense references make a step cross many boundaries and whole windows."""
    rng = np.random.default_rng(23)
    for m, lens, wkind in ((40_000, (50, 51), "small"), (600_000, (50, 51), "small"), (40_000, (20, 3000), "small"),
                           (40_000, (50, 51), "mixed"), (600_000, (30, 400), "signed")):
        refs = synth.genome_intervals(m, 31 + m % 5, 50, 2000)
        reads = synth.genome_intervals(500_000, 32, lens[0], lens[1])
        n = len(reads)
        if wkind == "small":
            w = rng.integers(0, 100, size=n)
        elif wkind == "signed":
            w = rng.integers(-1000, 1000, size=n)
        else:
            w = rng.integers(0, 50, size=n)
            big = rng.integers(0, n, size=200)
            w[big] = rng.integers(1 << 22, (1 << 31) - 1, size=200)            # steps holding one of these fall back
            w[rng.integers(0, n, size=50)] = -(1 << 30)
        check(engine, refs, reads, w.astype(np.int32), n_classes=synth.n_classes())


def first_violation(reads):
    bad = (reads[1:, 0] < reads[:-1, 0]) | ((reads[1:, 0] == reads[:-1, 0]) & (reads[1:, 1] < reads[:-1, 1]))
    nz = np.nonzero(bad)[0]
    return int(nz[0]) + 1 if len(nz) else -1


def test_check_sorted_without_the_sorted_hint(engine):
    """GTX_CHECK_SORTED alone (no GTX_READS_SORTED): the order is still verified -- the flag selects the streaming kernel,
    which is exact for any order -- on shuffled reads, on sorted reads, and over several batches of one stream."""
    rng = np.random.default_rng(91)
    refs = synth.genome_intervals(20000, 91, 50, 2000)
    reads = synth.genome_intervals(300000, 92, 50, 51)
    shuffled = reads[rng.permutation(len(reads))]
    engine.set_refs(refs, synth.n_classes())
    hits, info = engine.count(shuffled, None, gtx.CHECK_SORTED)
    np.testing.assert_array_equal(hits, orc.count(refs, shuffled, algo=orc.BIN_INDEX))
    assert info["first_unsorted"] == first_violation(shuffled)
    hits, info = engine.count(reads, None, gtx.CHECK_SORTED)
    np.testing.assert_array_equal(hits, orc.count(refs, reads, algo=orc.BIN_INDEX))
    assert info["first_unsorted"] == -1
    late = reads.copy()
    late[[250000, 250001]] = late[[250001, 250000]]
    if late[250000, 1] == late[250001, 1]:
        late[250001, 1] -= 1
    cuts = [0, 100000, 250001, len(reads)]
    _, info = engine.count_stream([(late[a:b], None) for a, b in zip(cuts[:-1], cuts[1:])], gtx.CHECK_SORTED)
    assert info["first_unsorted"] == 250001


def test_page_locked_source_and_many_batches():
    """Host buffers from gtx_host_alloc are read by the DMA engine directly (no staging copy), pageable ones go through
    the pinned slots; both are double-buffered over many small batches and must give the oracle's counts, info included."""
    import os
    os.environ["GTX_BATCH_READS"] = "30000"
    try:
        e = gtx.Engine(0)
    finally:
        del os.environ["GTX_BATCH_READS"]
    rng = np.random.default_rng(93)
    refs = synth.genome_intervals(8000, 93, 50, 3000)
    reads = synth.genome_intervals(400_003, 94, 30, 400)
    reads[123456, 0] = 99                                           # unknown class
    reads[300000, 2] = reads[300000, 1] - 5                         # degenerate
    w = rng.integers(0, 4, size=len(reads)).astype(np.int32)
    e.set_refs(refs, synth.n_classes())
    want = orc.count(refs, np.delete(reads, [300000], axis=0), np.delete(w, [300000]))
    pr = e.pinned_array(reads.shape); pr[:] = reads
    pw = e.pinned_array(w.shape); pw[:] = w
    for rr, ww in ((reads, w), (pr, pw), (pr, None), (reads, None)):
        for flags in (gtx.READS_SORTED, 0):
            hits, info = e.count(rr, ww, flags)
            ref = want if ww is not None else orc.count(refs, np.delete(reads, [300000], axis=0))
            np.testing.assert_array_equal(hits, ref)
            assert info["n_no_class"] == 1 and info["n_degenerate"] == 1 and info["first_degenerate"] == 300000
    # two page-locked producer buffers filled in turn (the contract of gtx_host_alloc): reuse after the NEXT call returned
    bufs = [e.pinned_array((50_000, 3)), e.pinned_array((50_000, 3))]
    lib, ctx = e.lib, e.ctx
    assert lib.gtx_count_begin(ctx) == 0
    for i, at in enumerate(range(0, len(reads), 50_000)):
        part = reads[at:at + 50_000]
        b = bufs[i & 1]
        b[:len(part)] = part
        assert lib.gtx_count_add(ctx, b.ctypes.data, None, len(part), gtx.READS_SORTED) == 0
    hits = np.zeros(len(refs), dtype=np.uint64)
    assert lib.gtx_count_end(ctx, hits.ctypes.data, None) == 0
    np.testing.assert_array_equal(hits, orc.count(refs, np.delete(reads, [300000], axis=0)))
    win, _ = e.scan(pr, synth.CHROM_LEN, 1000, 3000, weights=pw)
    valid = np.delete(np.arange(len(reads)), [123456, 300000])
    want_w, _ = orc.scan(reads[valid], synth.CHROM_LEN, 1000, 3000, weights=w[valid])
    np.testing.assert_array_equal(win, want_w)
    cov, _ = e.coverage(pr, pw)
    np.testing.assert_array_equal(cov, orc.coverage(refs, reads[valid], w[valid]))
    e.close()


def test_abandoned_call_leaves_no_stale_info(engine):
    """A count stream that is begun, fed and never ended must not leak its n_no_class / n_degenerate into the next call
    of either kind (count and coverage share the info block)."""
    refs = synth.genome_intervals(5000, 95, 50, 2000)
    reads = synth.genome_intervals(50000, 96, 50, 51)
    junk = reads.copy(); junk[:, 0] = 77                              # all of an unknown class
    engine.set_refs(refs, synth.n_classes())
    lib, ctx = engine.lib, engine.ctx
    assert lib.gtx_count_begin(ctx) == 0
    assert lib.gtx_count_add(ctx, junk.ctypes.data, None, len(junk), gtx.READS_SORTED) == 0
    cov, info = engine.coverage(reads)                                # the count stream above is abandoned
    assert info["n_no_class"] == 0 and info["n_degenerate"] == 0
    np.testing.assert_array_equal(cov, orc.coverage(refs, reads))
    assert lib.gtx_coverage_begin(ctx) == 0
    assert lib.gtx_coverage_add(ctx, junk.ctypes.data, None, len(junk), 0) == 0
    hits, info = engine.count(reads)                                  # ... and the other way round
    assert info["n_no_class"] == 0
    np.testing.assert_array_equal(hits, orc.count(refs, reads))


def test_extreme_coordinates(engine):
    """Coordinates at the edge of the packed representation (|x| up to 2^31-3) on every path: streaming kernel, per-read
    search kernel, bucket path (forced for a small batch)."""
    import os
    top = 2**31 - 3
    rng = np.random.default_rng(97)
    refs = synth.genome_intervals(30000, 97, 50, 2000)
    refs = np.concatenate([refs, np.array([[0, 1, top], [0, top - 5, top], [5, top, top], [23, 100, top]], dtype=np.int32)])
    reads = synth.genome_intervals(300000, 98, 50, 51)
    extra = np.array([[0, top, top], [0, 1, top], [0, top - 1, top], [5, 7, top], [23, top - 3, top - 2], [23, 248_000_000, top]], dtype=np.int32)
    reads = np.concatenate([reads, extra])
    reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    want = orc.count(refs, reads, algo=orc.BIN_INDEX)
    engine.set_refs(refs, synth.n_classes())
    for rr in (reads, reads[rng.permutation(len(reads))]):
        for flags in (gtx.READS_SORTED, 0):
            hits, info = engine.count(rr, None, flags)
            np.testing.assert_array_equal(hits, want)
            assert info["n_degenerate"] == 0
    os.environ["GTX_BUCKET_MIN_READS"] = "1"
    try:
        e = gtx.Engine(0)
    finally:
        del os.environ["GTX_BUCKET_MIN_READS"]
    e.set_refs(refs, synth.n_classes())
    hits, _ = e.count(reads[rng.permutation(len(reads))], None, 0)
    np.testing.assert_array_equal(hits, want)
    e.close()


def test_config5_1b_reads_2m_refs(engine):
    """BASELINE config 5, count part: 1 B reads (12 GB resident in HBM) x 2 M regions over the 24 hg38 chromosomes.
    The CPU oracle does not run at this size, so: (i) counts are additive over a split of the stream into uneven parts,
    (ii) the grand total equals the sum over the parts, (iii) one chromosome's regions are checked against the oracle on
    that chromosome's ~15 M reads, (iv) nothing was dropped (no degenerate / unknown-class reads)."""
    torch = pytest.importorskip("torch")
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = 1_000_000_000
    dev = torch.device("cuda", 0)
    reads = bench.make_reads_on_device(n, np.arange(24), 55, dev)
    refs = synth.genome_intervals(2_000_000, 45, 50, 2000)
    engine.set_refs(refs, 24)
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    whole = torch.zeros(len(refs), dtype=torch.int64, device=dev)
    part = torch.zeros_like(whole)
    acc = torch.zeros_like(whole)
    engine.count_device(reads.data_ptr(), n, whole.data_ptr())
    info = engine.last_info()
    assert info["n_degenerate"] == 0 and info["n_no_class"] == 0
    cuts = [0, 3, 333_333_333, 700_000_001, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        engine.count_device(reads[a:b].data_ptr(), b - a, part.data_ptr())
        acc += part
    engine.sync()
    engine.set_stream(0)
    assert torch.equal(whole, acc)
    c = int(np.argmin(synth.CHROM_LEN))                               # the shortest chromosome: ~1.5 % of the reads
    sub = reads[reads[:, 0] == c].cpu().numpy()
    assert 5_000_000 < len(sub) < 40_000_000
    rsel = np.nonzero(refs[:, 0] == c)[0]
    want = orc.count(refs[rsel], sub, algo=orc.SORTED_MERGE)
    np.testing.assert_array_equal(whole.cpu().numpy().view(np.uint64)[rsel], want)
    # every read overlaps the regions it overlaps: grand total = sum over reads of its overlap count, checked on the sample's class
    assert int(whole.sum().item()) > n // 2


def merge_inputs(rng, m, n, nc, span, inv_refs, inv_reads):
    def mk(k, inv_p):
        c = rng.integers(0, nc, size=k); s = rng.integers(-5, span, size=k)
        e = s + rng.integers(0, 400, size=k) - 1                     # zero-length ones included (e = s - 1)
        e = np.where(rng.random(k) < inv_p, s - rng.integers(2, 300, size=k), e)
        a = np.stack([c, s, e], 1).astype(np.int32)
        return a[np.lexsort((a[:, 1], a[:, 0]))]
    return mk(m, inv_refs), mk(n, inv_reads)


@pytest.mark.parametrize("m,n,inv_refs,inv_reads", [(3000, 200_000, 0.0, 0.001), (3000, 200_000, 0.01, 0.0), (5000, 300_000, 0.02, 0.01),
                                                    (40, 500, 0.3, 0.3), (2000, 600_000, 0.001, 0.0005)])
def test_sorted_merge_semantics_with_inverted_intervals(engine, m, n, inv_refs, inv_reads):
    """GTX_ZERO_LENGTH_OK on a GTX_REFS_KEEP_ZERO_LENGTH reference set = the merge's matching rule for ANY interval
    (genomic_intervals.cpp:1225-1236, :5844-5918): inverted reads and regions take part by the two comparisons.  Against the
    restated merge, on the streaming kernel, the per-read search kernel and the bucket path, weighted and not, in one call
    and over several batches."""
    rng = np.random.default_rng(m + n)
    refs, reads = merge_inputs(rng, m, n, 5, 3_000_000, inv_refs, inv_reads)
    w = rng.integers(-2, 6, size=n).astype(np.int32)
    engine.set_refs(refs, 5, gtx.REFS_KEEP_ZERO_LENGTH)
    n_inv = int((reads[:, 1] > reads[:, 2] + 1).sum())
    for ww in (None, w):
        want = orc.count(refs, reads, ww, algo=orc.SORTED_MERGE)
        for flags in (gtx.READS_SORTED, 0):
            hits, info = engine.count(reads, ww, flags | gtx.ZERO_LENGTH_OK)
            np.testing.assert_array_equal(hits, want)
            assert info["n_degenerate"] == n_inv and info["n_unplaced"] == 0
        cuts = [0, n // 3, n // 3 + 1, n]
        hits, _ = engine.count_stream([(reads[a:b], None if ww is None else ww[a:b]) for a, b in zip(cuts[:-1], cuts[1:])],
                                      gtx.READS_SORTED | gtx.ZERO_LENGTH_OK)
        np.testing.assert_array_equal(hits, want)
        cov, _ = engine.coverage(reads, ww, gtx.ZERO_LENGTH_OK)                      # clamped overlaps: inverted intervals add nothing
        np.testing.assert_array_equal(cov, orc.coverage(refs, reads, ww, algo=orc.SORTED_MERGE))
    # the call after one with inverted reads starts clean
    plain = reads[reads[:, 1] <= reads[:, 2] + 1]
    hits, info = engine.count(plain, None, gtx.READS_SORTED | gtx.ZERO_LENGTH_OK)
    np.testing.assert_array_equal(hits, orc.count(refs, plain, algo=orc.SORTED_MERGE))
    assert info["n_degenerate"] == 0


def test_gaps_coverage_formula_with_inverted_intervals(engine):
    """CalcIndexCoverage under match_gaps adds min(stops) - max(starts) + 1 per match WITHOUT clamping at 0
    (genomic_intervals.cpp:5278): pairs with an inverted interval subtract.  GTX_GAPS_FORMULA against a direct evaluation."""
    rng = np.random.default_rng(123)
    refs, reads = merge_inputs(rng, 800, 60_000, 3, 400_000, 0.02, 0.01)
    w = rng.integers(0, 5, size=len(reads)).astype(np.int32)
    engine.set_refs(refs, 3, gtx.REFS_KEEP_ZERO_LENGTH)
    cov, info = engine.coverage(reads, w, gtx.ZERO_LENGTH_OK | gtx.GAPS_FORMULA)
    want = np.zeros(len(refs), dtype=np.uint64)
    for k in range(len(refs)):
        c, rs, re = map(int, refs[k])
        sel = (reads[:, 0] == c) & (reads[:, 1] <= re) & (reads[:, 2] >= rs)
        ov = np.minimum(reads[sel, 2], re).astype(np.int64) - np.maximum(reads[sel, 1], rs) + 1
        want[k] = np.uint64(int((ov * w[sel]).sum()) % (1 << 64))
    np.testing.assert_array_equal(cov, want)
    assert info["n_unplaced"] == 0


def test_placeholder_regions_never_match(engine):
    refs = np.array([[0, 100, 200], [-1, 1, 0], [0, 150, 300], [-1, 120, 130]], dtype=np.int32)
    reads = np.array([[0, 90, 160], [0, 125, 126], [0, 290, 400]], dtype=np.int32)
    for fl in (0, gtx.REFS_KEEP_ZERO_LENGTH):
        engine.set_refs(refs, 1, fl)
        hits, _ = engine.count(reads, None, gtx.READS_SORTED | (gtx.ZERO_LENGTH_OK if fl else 0))
        assert hits.tolist() == [2, 0, 2, 0]


@pytest.mark.gpu
def test_slot_widths_take_turns_on_one_context(engine):
    """An unweighted *_device call of one launch counts into 32-bit histogram slots (its finalize step reads the same buffers as
    unsigned[]); weighted calls, reads in no order, host-buffer streams and coverage keep 64-bit slots.  The two views share the
    buffers and each leaves its own cleared: any sequence of calls on one context must give every call's own result -- including
    a dense reference set (the all-boundaries-at-once kernel), the order check, reads of no class and a batch small enough to keep
    the tile sums in the streaming kernel."""
    import torch
    rng = np.random.default_rng(41)
    for m, n in ((30_000, 1_500_000), (400_000, 600_000), (5_000, 90_000)):          # sparse, dense (flip), small batch
        refs = synth.genome_intervals(m, 3 + m, 50, 2000)
        reads = synth.genome_intervals(n, 4 + m, 50, 300)
        reads[7, 0] = 999                                                        # no such class
        w = rng.integers(-3, 9, size=n).astype(np.int32)
        engine.set_refs(refs, synth.n_classes())
        d_r, d_w = torch.from_numpy(reads).cuda(), torch.from_numpy(w).cuda()
        shuf = reads[rng.permutation(n)]
        d_s = torch.from_numpy(np.ascontiguousarray(shuf)).cuda()
        hits = torch.zeros(m, dtype=torch.int64, device="cuda")
        keep = np.ones(n, dtype=bool); keep[7] = False
        want = orc.count(refs, reads[keep], algo=orc.SORTED_MERGE)
        want_w = orc.count(refs, reads[keep], w[keep], algo=orc.SORTED_MERGE)
        def got():
            engine.sync()
            return hits.cpu().numpy().view(np.uint64)
        for step in ("u32", "u64w", "u32", "shuffled", "u32check", "host", "cover", "u32", "u64w"):
            hits.fill_(-1)
            if step == "u32":
                engine.count_device(d_r.data_ptr(), n, hits.data_ptr()); np.testing.assert_array_equal(got(), want)
                assert engine.last_info()["n_no_class"] == 1
            elif step == "u32check":
                engine.count_device(d_r.data_ptr(), n, hits.data_ptr(), flags=gtx.READS_SORTED | gtx.CHECK_SORTED); np.testing.assert_array_equal(got(), want)
                assert engine.last_info()["first_unsorted"] == 8                     # (the read of no class sits out of order: the one behind it is reported)
            elif step == "u64w":
                engine.count_device(d_r.data_ptr(), n, hits.data_ptr(), d_weights=d_w.data_ptr()); np.testing.assert_array_equal(got(), want_w)
            elif step == "shuffled":
                engine.count_device(d_s.data_ptr(), n, hits.data_ptr(), flags=0); np.testing.assert_array_equal(got(), want)
            elif step == "host":
                h, _ = engine.count(reads, None); np.testing.assert_array_equal(h, want)
            else:
                c, _ = engine.coverage(reads[keep][:200_000]); np.testing.assert_array_equal(c, orc.coverage(refs, reads[keep][:200_000], algo=orc.SORTED_MERGE))
