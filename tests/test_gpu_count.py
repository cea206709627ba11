"""Parity of the HIP count path (through the C ABI) against the CPU oracle -- bit-exact (uint64)."""
import numpy as np
import pytest

import gtx
from gtx import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def both_oracles(refs, reads, weights=None):
    a = orc.count(refs, reads, weights, algo=orc.BIN_INDEX)
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
