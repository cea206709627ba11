"""The bucket path for reads in no particular order (gtx_bucket.hip: partition by position bucket, count in LDS)
against the CPU oracle.  GTX_BUCKET_MIN_READS=1 sends every unsorted call through it, however small."""
import os

import numpy as np
import pytest

import gtx
from gtx import synth
from oracle import orc
from test_gpu_fuzz import gen, orders

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def beng():
    os.environ["GTX_BUCKET_MIN_READS"] = "1"
    try:
        e = gtx.Engine(0)
    finally:
        del os.environ["GTX_BUCKET_MIN_READS"]
    yield e
    e.close()


def check(e, refs, reads, n_classes, w=None, flags=0):
    e.set_refs(refs, n_classes)
    got, info = e.count(reads, w, flags)
    sel = reads[:, 0] < n_classes
    want = orc.count(refs, reads[sel], None if w is None else w[sel], algo=orc.BIN_INDEX)
    np.testing.assert_array_equal(got, want)
    return info


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_small_tables(beng, seed):
    rng = np.random.default_rng(7000 + seed)
    for _ in range(10):
        n = int(rng.choice([0, 1, 63, 257, 1000, 5000]))
        m = int(rng.choice([0, 1, 2, 64, 500, 3000, 9000]))
        n_classes = int(rng.choice([1, 2, 5, 40]))
        span = int(rng.choice([10, 300, 100000]))
        refs, reads = gen(rng, n, m, n_classes, span, allow_invalid_refs=True)
        for r in (orders(rng, reads) if n else [reads]):
            info = check(beng, refs, r, n_classes)
            assert info["n_no_class"] == int((reads[:, 0] >= n_classes).sum())
        if n:
            w = rng.integers(-2, 6, size=n).astype(np.int32)
            check(beng, refs, reads, n_classes, w)


def test_many_buckets_per_class_shuffled(beng):
    # 300k regions on 24 chromosomes: ~160 buckets; 2M shuffled reads, one chromosome without regions
    rng = np.random.default_rng(5)
    refs = synth.genome_intervals(300_000, 5, 50, 2000)
    refs = refs[refs[:, 0] != 7]
    reads = synth.genome_intervals(2_000_000, 6, 36, 36)
    reads = reads[rng.permutation(len(reads))]
    check(beng, refs, reads, synth.n_classes())
    w = rng.integers(0, 4, size=len(reads)).astype(np.int32)
    check(beng, refs, reads, synth.n_classes(), w)


def test_reads_longer_than_a_bucket_fall_back_for_the_starts_histogram(beng):
    # dense short regions (a bucket spans ~40 kb) and reads of up to 5 Mb: their ends lie far beyond the LDS slice
    rng = np.random.default_rng(6)
    m = 60_000
    rs = np.sort(rng.integers(1, 1_200_000, size=m))
    refs = np.stack([np.zeros(m, dtype=np.int64), rs, rs + rng.integers(0, 30, size=m)], axis=1).astype(np.int32)
    n = 200_000
    qs = rng.integers(1, 1_200_000, size=n)
    ql = rng.choice([10, 1000, 200_000, 5_000_000], size=n)
    reads = np.stack([np.zeros(n, dtype=np.int64), qs, qs + ql], axis=1).astype(np.int32)
    check(beng, refs, reads, 1)


def test_duplicate_boundaries_across_bucket_cuts(beng):
    # 10k regions that all end at one of 3 positions: equal boundaries straddle every bucket cut
    rng = np.random.default_rng(8)
    m = 10_000
    ends = rng.choice([1000, 2000, 3000], size=m)
    refs = np.stack([np.zeros(m, dtype=np.int64), ends - rng.integers(0, 900, size=m), ends], axis=1).astype(np.int32)
    n = 50_000
    qs = rng.choice([999, 1000, 1001, 1999, 2000, 2001, 2999, 3000, 3001, 1, 5000], size=n)
    reads = np.stack([np.zeros(n, dtype=np.int64), qs, qs + rng.integers(0, 3, size=n)], axis=1).astype(np.int32)
    check(beng, refs, reads, 1)


def test_sorted_semantics_flags_and_degenerate_reads(beng):
    rng = np.random.default_rng(9)
    refs = synth.refs_single_chrom(5000, seed=3, chrom_len=1_000_000)
    reads = synth.reads_single_chrom(20000, seed=4, chrom_len=1_000_000)
    reads[::50, 2] = reads[::50, 1] - 1                        # zero-length
    reads[::333, 2] = reads[::333, 1] - 5                      # inverted: reported, never counted
    reads = reads[rng.permutation(len(reads))]
    beng.set_refs(refs)
    got, info = beng.count(reads, None, 0)
    ok = reads[:, 1] <= reads[:, 2]
    np.testing.assert_array_equal(got, orc.count(refs, reads[ok], algo=orc.BIN_INDEX))
    assert info["n_degenerate"] == int((~ok).sum())
    got, info = beng.count(reads, None, gtx.ZERO_LENGTH_OK)
    assert info["n_degenerate"] == int((reads[:, 1] > reads[:, 2] + 1).sum())


def test_streamed_unsorted_batches_add_up(beng):
    rng = np.random.default_rng(10)
    refs = synth.genome_intervals(40_000, 11, 50, 3000)
    reads = synth.genome_intervals(300_000, 12, 30, 200)
    reads = reads[rng.permutation(len(reads))]
    beng.set_refs(refs, synth.n_classes())
    got, _ = beng.count_stream([(b, None) for b in np.array_split(reads, 7)], 0)
    np.testing.assert_array_equal(got, orc.count(refs, reads, algo=orc.BIN_INDEX))


def test_thousands_of_buckets_and_extreme_ends(beng):
    """5 M regions: ~2500 buckets (the split kernel's per-bucket LDS tables grow, its tile shrinks), weighted and not; reads that
    end at the top of the coordinate range (rank = everything: no padding value may be mistaken for a boundary)."""
    rng = np.random.default_rng(11)
    refs = synth.genome_intervals(5_000_000, 12, 50, 1500)
    reads = synth.genome_intervals(600_000, 13, 30, 5000)
    top = 2**31 - 3
    reads[rng.integers(0, len(reads), size=50), 2] = top
    reads = reads[rng.permutation(len(reads))]
    w = rng.integers(0, 4, size=len(reads)).astype(np.int32)
    check(beng, refs, reads, synth.n_classes())
    check(beng, refs, reads, synth.n_classes(), w)


def test_reads_in_runs_and_in_stripes(beng):
    # the scatter pass deals every block's arena out in chunks of 64 pairs per bucket (bucket_scatter_kernel): reads that alternate
    # between two chromosomes in stripes of 64, and the same reads sorted (a bucket's reads are one stretch of the stream: whole tiles
    # go to one bucket, 64 new chunks at a time)
    rng = np.random.default_rng(77)
    refs = synth.genome_intervals(200_000, 9, 50, 2000)
    refs = refs[refs[:, 0] < 2]
    n = 64 * 16 * 400
    unit = np.arange(n) // 64
    cls = np.where(unit % 16 == 0, 0, 1)
    qs = rng.integers(1, 150_000_000, size=n)
    reads = np.stack([cls, qs, qs + 35], axis=1).astype(np.int32)
    check(beng, refs, reads, 2)
    w = rng.integers(-1, 5, size=n).astype(np.int32)
    check(beng, refs, reads, 2, w)
    srt = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    check(beng, refs, srt, 2)
    check(beng, refs, srt, 2, w)


def test_more_classes_than_the_lookup_tables_take(beng):
    # 3000 classes: the scatter kernel's LDS tables (16 B per class) do not take them and the per-read search kernel serves
    rng = np.random.default_rng(91)
    n_classes, m, n = 3000, 20_000, 50_000
    rc = np.sort(rng.integers(0, n_classes, size=m))
    rs = rng.integers(1, 100_000, size=m)
    refs = np.stack([rc, rs, rs + rng.integers(0, 500, size=m)], axis=1).astype(np.int32)
    refs = refs[np.lexsort((refs[:, 1], refs[:, 0]))]
    qs = rng.integers(1, 100_000, size=n)
    reads = np.stack([rng.integers(0, n_classes + 5, size=n), qs, qs + rng.integers(0, 300, size=n)], axis=1).astype(np.int32)
    check(beng, refs, reads, n_classes)


# ---- coverage through the partition path (bucket_cover_kernel): reads in no particular order, flags without the sorted hint ----
def check_cov(e, refs, reads, n_classes, w=None, flags=0):
    e.set_refs(refs, n_classes)
    got, info = e.coverage(reads, w, flags)
    sel = reads[:, 0] < n_classes
    want = orc.coverage(refs, reads[sel], None if w is None else w[sel], algo=orc.BIN_INDEX)
    np.testing.assert_array_equal(got, want)
    return info


@pytest.mark.parametrize("seed", range(6))
def test_coverage_fuzz_small_tables(beng, seed):
    rng = np.random.default_rng(9100 + seed)
    for _ in range(8):
        n = int(rng.choice([1, 63, 257, 1000, 5000]))
        m = int(rng.choice([0, 1, 2, 64, 500, 3000, 9000]))
        n_classes = int(rng.choice([1, 2, 5, 40]))
        span = int(rng.choice([10, 300, 100000]))
        refs, reads = gen(rng, n, m, n_classes, span, allow_invalid_refs=True)
        reads = reads[reads[:, 1] <= reads[:, 2]]                    # (inverted reads are the merge's business: test_gpu_count / CLI tests)
        if len(reads) == 0:
            continue
        for r in orders(rng, reads):
            check_cov(beng, refs, r, n_classes, flags=gtx.READS_UNSORTED)
        w = rng.integers(-2, 6, size=len(reads)).astype(np.int32)
        p = rng.permutation(len(reads))
        check_cov(beng, refs, reads[p], n_classes, w[p], flags=gtx.READS_UNSORTED)


def test_coverage_shuffled_large(beng):
    # 300 k regions (600 k thresholds: ~300 buckets), 2 M shuffled reads of 20-4000 bp: ends beyond the slice, one chromosome without
    # regions, the host-side sample decides (flags = 0), and the same call again in sorted order through the streaming kernel
    rng = np.random.default_rng(15)
    refs = synth.genome_intervals(300_000, 5, 50, 2000)
    refs = refs[refs[:, 0] != 7]
    reads = synth.genome_intervals(2_000_000, 6, 20, 4000)
    shuf = reads[rng.permutation(len(reads))]
    check_cov(beng, refs, shuf, synth.n_classes())
    w = rng.integers(0, 4, size=len(reads)).astype(np.int32)
    check_cov(beng, refs, shuf, synth.n_classes(), w)
    check_cov(beng, refs, reads, synth.n_classes(), w, flags=gtx.READS_SORTED)
    # count and coverage in turn on one context: neither disturbs the other's zeroed state
    check(beng, refs, shuf, synth.n_classes())
    check_cov(beng, refs, shuf, synth.n_classes())


# ---- genomic_scans counts through the partition path (bucket_scanhist_kernel): unsorted rule, start positions ----
SCAN_LENS = synth.CHROM_LEN // 20


def scan_reads(n, seed):
    r = synth.genome_intervals(n, seed, 50, 51)
    r[:, 1] = r[:, 1] // 20 + 1
    r[:, 2] = r[:, 1] + 49
    return r


@pytest.mark.parametrize("step,size", [(1000, 1000), (25, 500), (100, 300), (7, 7 * 13), (1, 4), (25, 1600), (10, 1000)])   # (the last: more micro-windows per window than the parts sum themselves)
def test_scan_shuffled_reads(beng, step, size):
    rng = np.random.default_rng(61)
    reads = scan_reads(300_000, 61)
    reads = reads[rng.permutation(len(reads))]
    # rows the unsorted scanner ignores: start > stop, stop <= 0, start < 1, beyond the chromosome, unknown class
    odd = np.array([[0, 500, 400], [1, -30, 0], [2, -5, 20], [3, 0, 10], [4, int(SCAN_LENS[4]) + 5000, int(SCAN_LENS[4]) + 5050], [40, 10, 60],
                    [5, int(SCAN_LENS[5]), int(SCAN_LENS[5]) + 10], [6, 1, 1]], dtype=np.int32)
    reads = np.concatenate([reads[:1000], odd, reads[1000:]])
    for flags in (gtx.READS_UNSORTED, 0):                           # the caller's word / the library's own sample of the host buffer
        got, off = beng.scan(reads, SCAN_LENS, step, size, "1", flags=flags)
        want, woff = orc.scan(reads, SCAN_LENS, step, size, "1", algo=0)
        np.testing.assert_array_equal(off, woff)
        np.testing.assert_array_equal(got, want)
    w = rng.integers(-3, 6, size=len(reads)).astype(np.int32)
    got, _ = beng.scan(reads, SCAN_LENS, step, size, "1", weights=w, flags=gtx.READS_UNSORTED)
    want, _ = orc.scan(reads, SCAN_LENS, step, size, "1", weights=w, algo=0)
    np.testing.assert_array_equal(got, want)


def test_scan_shuffled_reads_centres_and_the_sorted_rule(beng):
    # preprocess 'c': placed by the centre (reads of 1..4000 bp, some starting below 1 with their centre inside, some with the centre
    # beyond the chromosome); the sorted scanner's rule is not the partition path's: the general kernels
    rng = np.random.default_rng(62)
    reads = scan_reads(100_000, 62)
    reads[:, 2] = reads[:, 1] + rng.integers(0, 4000, size=len(reads))
    reads[:50, 1] -= 3000
    reads = reads[rng.permutation(len(reads))]
    for step, size in ((200, 1000), (25, 500), (1000, 1000)):
        got, _ = beng.scan(reads, SCAN_LENS, step, size, "c", flags=gtx.READS_UNSORTED)
        want, _ = orc.scan(reads, SCAN_LENS, step, size, "c", algo=0)
        np.testing.assert_array_equal(got, want)
    w = rng.integers(-3, 6, size=len(reads)).astype(np.int32)
    got, _ = beng.scan(reads, SCAN_LENS, 200, 1000, "c", weights=w)
    want, _ = orc.scan(reads, SCAN_LENS, 200, 1000, "c", weights=w, algo=0)
    np.testing.assert_array_equal(got, want)
    srt = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    got, _ = beng.scan(srt, SCAN_LENS, 200, 1000, "1", flags=gtx.ZERO_LENGTH_OK | gtx.READS_UNSORTED)
    want, _ = orc.scan(srt, SCAN_LENS, 200, 1000, "1", algo=1)
    np.testing.assert_array_equal(got, want)
