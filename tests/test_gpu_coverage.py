"""Parity of the HIP coverage path (gtx_coverage*) against the CPU oracle's CalcIndexCoverage restatement -- bit-exact."""
import numpy as np
import pytest

import gtx
from gtx import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def check(engine, refs, reads, weights=None, n_classes=0, algo=orc.BIN_INDEX):
    engine.set_refs(refs, n_classes)
    got, info = engine.coverage(reads, weights)
    want = orc.coverage(refs, reads, weights, algo=algo)
    np.testing.assert_array_equal(got, want)
    return got


def test_toy(engine):
    # G2 toy with -i: A 2 / B 102 / C 1 (by hand: r2 1, r3 1+51, r4 50, r5 1, r6 1)
    refs = np.array([[0, 101, 200], [0, 151, 300], [1, 11, 20]], dtype=np.int32)
    reads = np.array([[0, 51, 100], [0, 51, 101], [0, 200, 250], [0, 201, 250], [0, 300, 350], [1, 20, 30], [2, 2, 5]], dtype=np.int32)
    assert check(engine, refs, reads).tolist() == [2, 102, 1]


@pytest.mark.parametrize("n,m,seed", [(1, 1, 0), (64, 64, 2), (4097, 300, 4), (200000, 20000, 5)])
def test_single_chrom(engine, n, m, seed):
    refs = synth.refs_single_chrom(m, seed=seed, chrom_len=3_000_000, max_len=5000)
    rng = np.random.default_rng(seed)
    reads = synth.reads_single_chrom(n, seed=seed, chrom_len=3_000_000)
    reads[:, 2] = reads[:, 1] + rng.integers(0, 3000, size=n)       # variable read lengths: every containment case occurs
    check(engine, refs, reads)
    check(engine, refs, reads, algo=orc.SORTED_MERGE)


def test_multi_chrom_weights_and_unsorted(engine):
    rng = np.random.default_rng(7)
    refs = synth.genome_intervals(20000, 7, 50, 8000)
    reads = synth.genome_intervals(300000, 8, 20, 4000)
    check(engine, refs, reads, n_classes=24)
    w = rng.integers(-2, 5, size=len(reads)).astype(np.int32)
    check(engine, refs, reads, w, n_classes=24)
    perm = rng.permutation(len(reads))
    check(engine, refs, reads[perm], w[perm], n_classes=24)


@pytest.mark.parametrize("run", [64, 256, 300, 1000, 5000])
def test_reads_of_one_length_in_runs(engine, run):
    """sorted reads whose length is constant over runs of `run` reads and changes between them -- 1 bp, 36, 150, 5000, 100 kb, 1 Mb:
    a step of 256 reads all of one length lets the second window of the streaming kernel search the keys the first one staged
    (the ends are the starts plus a constant), a step that straddles two runs does not; dense and sparse reference sets, against both
    restated algorithms"""
    rng = np.random.default_rng(run)
    n = 60_000
    starts = np.sort(rng.integers(1, 40_000_000, size=n))
    lens = np.repeat(rng.choice([1, 36, 150, 5000, 100_000, 1_000_000], size=n // run + 1), run)[:n]
    reads = np.stack([np.zeros(n, dtype=np.int64), starts, starts + lens - 1], axis=1).astype(np.int32)
    for m in (300, 60_000):
        refs = synth.refs_single_chrom(m, seed=run + m, chrom_len=41_000_000, max_len=3000)
        check(engine, refs, reads, n_classes=1)
        check(engine, refs, reads, n_classes=1, algo=orc.SORTED_MERGE)
        check(engine, refs, reads, weights=rng.integers(0, 9, size=n).astype(np.int32), n_classes=1)      # (the weighted step does the same)


def test_regions_that_never_count(engine):
    refs = np.array([[0, 100, 200], [0, 300, 250], [0, 151, 150], [0, -5, 0]], dtype=np.int32)
    reads = np.array([[0, 1, 1000], [0, 150, 150], [0, 140, 160]], dtype=np.int32)
    assert check(engine, refs, reads).tolist() == [101 + 1 + 21, 0, 0, 0]


def test_count_and_coverage_interleave(engine):
    """the two reductions share the context: neither may disturb the other's zeroed state"""
    refs = synth.genome_intervals(5000, 9, 50, 3000)
    reads = synth.genome_intervals(100000, 10, 30, 500)
    engine.set_refs(refs, 24)
    c1, _ = engine.count(reads)
    v1, _ = engine.coverage(reads)
    c2, _ = engine.count(reads)
    v2, _ = engine.coverage(reads)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_array_equal(c1, orc.count(refs, reads))
    np.testing.assert_array_equal(v1, orc.coverage(refs, reads))


def test_weighted_sorted_reads_step_path_and_fallbacks(engine):
    """Weighted sorted reads take steps of 256 with the prefix sums of w and w x key in LDS (64 bits: any label value);
    variable read lengths break the order of the ends (general code for those steps); dense regions make a step cross windows."""
    rng = np.random.default_rng(71)
    for m, lens, wkind in ((30_000, (50, 51), "small"), (500_000, (50, 51), "small"), (30_000, (20, 2500), "small"),
                           (30_000, (50, 51), "huge"), (500_000, (36, 37), "signed")):
        refs = synth.genome_intervals(m, 70 + m % 3, 50, 2000)
        reads = synth.genome_intervals(400_000, 72, lens[0], lens[1])
        n = len(reads)
        w = {"small": rng.integers(0, 100, size=n), "signed": rng.integers(-1000, 1000, size=n),
             "huge": rng.integers(-(1 << 31), (1 << 31) - 1, size=n)}[wkind].astype(np.int32)
        check(engine, refs, reads, w, n_classes=synth.n_classes())
