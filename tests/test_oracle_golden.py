"""CPU suite, part 1: the oracle against the known-answer vectors of tests/golden/manifest.json and
against itself (its two independently restated algorithms must agree, as the reference's do)."""
import json
import os

import numpy as np
import pytest

from gtx import synth
from oracle import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]


def run_cli(cli, case):
    stdin = open(os.path.join(GOLD, case["stdin_file"]), "rb").read() if "stdin_file" in case else None
    cwd = os.getcwd()
    os.chdir(GOLD)
    try:
        return cli(case["args"], stdin)
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_known_answers(case):
    rc, out, err = run_cli(orc.cli, case)
    assert rc == case["rc"], err
    assert out == case["stdout"]
    assert case.get("stderr_contains", "") in err


def brute(refs, reads, weights=None):
    """Closed form on sorted copies: hits = sum w[s<=E] - sum w[e<S] per class (valid regions only)."""
    hits = np.zeros(len(refs), dtype=np.int64)
    w = np.ones(len(reads), dtype=np.int64) if weights is None else weights.astype(np.int64)
    for c in np.unique(refs[:, 0]):
        ri = np.nonzero(refs[:, 0] == c)[0]
        q = reads[:, 0] == c
        s, e, ww = reads[q, 1], reads[q, 2], w[q]
        os_, oe = np.argsort(s, kind="stable"), np.argsort(e, kind="stable")
        cs = np.concatenate([[0], np.cumsum(ww[os_])])
        ce = np.concatenate([[0], np.cumsum(ww[oe])])
        a = cs[np.searchsorted(s[os_], refs[ri, 2], side="right")]
        b = ce[np.searchsorted(e[oe], refs[ri, 1], side="left")]
        hits[ri] = a - b
    return hits.astype(np.uint64)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bin_index_equals_sorted_merge_equals_closed_form(seed):
    refs = synth.genome_intervals(5000, seed, 50, 3000)
    reads = synth.genome_intervals(60000, seed + 100, 30, 400)
    a = orc.count(refs, reads, algo=orc.BIN_INDEX)
    b = orc.count(refs, reads, algo=orc.SORTED_MERGE)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, brute(refs, reads))


def test_weights_and_unsorted_reads():
    rng = np.random.default_rng(5)
    refs = synth.genome_intervals(3000, 5, 50, 3000)
    reads = synth.genome_intervals(40000, 6, 30, 400)
    reads = reads[rng.permutation(len(reads))]
    w = rng.integers(0, 5, size=len(reads)).astype(np.int32)
    a = orc.count(refs, reads, w, algo=orc.BIN_INDEX)
    np.testing.assert_array_equal(a, brute(refs, reads, w))
    # max_label_value clamps the label (genomic_intervals.cpp:1081-1085)
    a3 = orc.count(refs, reads, w, algo=orc.BIN_INDEX, max_label_value=3)
    np.testing.assert_array_equal(a3, brute(refs, reads, np.minimum(w, 3)))


def test_sorted_merge_rejects_unsorted_queries():
    refs = synth.refs_single_chrom(100, seed=7, chrom_len=100000)
    reads = synth.reads_single_chrom(1000, seed=7, chrom_len=100000)[::-1].copy()
    with pytest.raises(orc.OracleError, match="query regions are not sorted"):
        orc.count(refs, reads, algo=orc.SORTED_MERGE)


def test_bin_index_rejects_degenerate_read():
    refs = np.array([[0, 100, 200]], dtype=np.int32)
    reads = np.array([[0, 151, 150]], dtype=np.int32)
    with pytest.raises(orc.OracleError, match="start position cannot be greater"):
        orc.count(refs, reads, algo=orc.BIN_INDEX)
    # ... but only on a chromosome the index knows (genomic_intervals.cpp:5719-5720)
    reads[0, 0] = 3
    assert orc.count(refs, reads, algo=orc.BIN_INDEX).tolist() == [0]


@pytest.mark.parametrize("step,size", [(1000, 1000), (25, 500), (100, 300)])
def test_scanners_agree(step, size):
    reads = synth.genome_intervals(50000, 9, 50, 51)
    a, off = orc.scan(reads, synth.CHROM_LEN // 100, step, size, algo=0)
    b, _ = orc.scan(reads, synth.CHROM_LEN // 100, step, size, algo=1)
    np.testing.assert_array_equal(a, b)
    # micro-window histogram + sliding sum by numpy
    comb = size // step
    for c, ln in enumerate(synth.CHROM_LEN // 100):
        n = int(ln) // step
        q = reads[reads[:, 0] == c]
        pos = q[:, 1].astype(np.int64)
        mw = (pos - 1) // step
        v = np.bincount(mw[(pos >= 1) & (mw < n)], minlength=n)[:n] if n else np.zeros(0, dtype=np.int64)
        nw = max(0, n - comb + 1)
        want = np.array([v[k:k + comb].sum() for k in range(nw)], dtype=np.uint64)
        np.testing.assert_array_equal(a[off[c]:off[c] + nw], want)


def test_sorted_merge_is_the_two_sided_condition_for_any_interval():
    """The restated merge (LoadIndexBuffer / GetMatch loop, genomic_intervals.cpp:5844-5918) on start-sorted inputs matches
    a query q and an index region r iff q.start <= r.stop and q.stop >= r.start -- whatever the order of start and stop
    inside either interval (zero-length and inverted ones included).  This is the closed form the device path implements
    (rank difference + pair kernels); coverage adds max(0, min(stops) - max(starts) + 1) per match."""
    rng = np.random.default_rng(5)
    for _ in range(400):
        nc = int(rng.integers(1, 4)); m = int(rng.integers(0, 25)); n = int(rng.integers(0, 40)); span = int(rng.integers(5, 60))

        def mk(k, inv_p):
            c = rng.integers(0, nc, size=k); s = rng.integers(-3, span, size=k)
            e = s + rng.integers(0, 15, size=k) - 1
            e = np.where(rng.random(k) < inv_p, s - rng.integers(1, 8, size=k), e)
            a = np.stack([c, s, e], 1).astype(np.int32)
            return a[np.lexsort((a[:, 1], a[:, 0]))]
        refs, reads = mk(m, rng.choice([0, 0.2, 0.5])), mk(n, rng.choice([0, 0.2, 0.5]))
        w = rng.integers(-2, 5, size=n).astype(np.int32)
        cnt, cov = np.zeros(m, dtype=np.uint64), np.zeros(m, dtype=np.uint64)
        for k in range(m):
            c, rs, re = map(int, refs[k])
            sel = (reads[:, 0] == c) & (reads[:, 1] <= re) & (reads[:, 2] >= rs)
            ov = np.maximum(np.minimum(reads[sel, 2], re).astype(np.int64) - np.maximum(reads[sel, 1], rs) + 1, 0)
            cnt[k] = np.uint64(int(w[sel].sum()) % (1 << 64)); cov[k] = np.uint64(int((ov * w[sel]).sum()) % (1 << 64))
        np.testing.assert_array_equal(orc.count(refs, reads, w, algo=orc.SORTED_MERGE), cnt)
        np.testing.assert_array_equal(orc.coverage(refs, reads, w, algo=orc.SORTED_MERGE), cov)
