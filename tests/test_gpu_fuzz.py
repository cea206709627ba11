"""Differential fuzzing of the HIP paths against the CPU oracle on many small adversarial inputs:
duplicate coordinates, nested / identical regions, negative and zero coordinates, empty classes,
reads clustered on boundaries, every order (sorted, reversed, shuffled, strand-interleaved)."""
import numpy as np
import pytest

import gtx
from oracle import orc

pytestmark = pytest.mark.gpu


def gen(rng, n, m, n_classes, span, allow_invalid_refs):
    rc = rng.integers(0, n_classes, size=m)
    rs = rng.integers(-5, span, size=m)
    rl = rng.choice([0, 0, 1, 2, 5, 50, span // 2], size=m) + rng.integers(0, 3, size=m)
    refs = np.stack([rc, rs, rs + rl], axis=1)
    if allow_invalid_refs and m:
        bad = rng.random(m) < 0.1
        refs[bad, 2] = refs[bad, 1] - rng.integers(1, 4, size=int(bad.sum()))       # start > end: never counts
    qc = rng.integers(0, n_classes + 1, size=n)                                     # one class beyond: ignored
    anchor = refs[rng.integers(0, max(m, 1), size=n), 1 + rng.integers(0, 2, size=n)] if m else rng.integers(0, span, size=n)
    qs = np.where(rng.random(n) < 0.5, anchor + rng.integers(-2, 3, size=n), rng.integers(1, span, size=n))
    qs = np.maximum(qs, 1)
    ql = rng.choice([0, 0, 1, 3, 49, span], size=n)
    reads = np.stack([qc, qs, qs + ql], axis=1)
    return refs.astype(np.int32), reads.astype(np.int32)


def orders(rng, reads):
    yield reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    yield reads[np.lexsort((reads[:, 1], reads[:, 0]))][::-1].copy()
    yield reads[rng.permutation(len(reads))]
    yield reads[np.argsort(reads[:, 1], kind="stable")]                             # position-sorted, classes interleaved


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_count_coverage_scan(engine, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(12):
        n = int(rng.choice([0, 1, 63, 64, 65, 255, 256, 257, 1000, 5000]))
        m = int(rng.choice([0, 1, 2, 62, 63, 64, 65, 127, 500, 3000]))
        n_classes = int(rng.choice([1, 2, 5, 40]))
        span = int(rng.choice([10, 300, 100000]))
        refs, reads = gen(rng, n, m, n_classes, span, allow_invalid_refs=True)
        engine.set_refs(refs, n_classes)
        want_c = orc.count(refs, reads[reads[:, 0] < n_classes], algo=orc.BIN_INDEX)
        want_v = orc.coverage(refs, reads[reads[:, 0] < n_classes], algo=orc.BIN_INDEX)
        w = rng.integers(-2, 6, size=n).astype(np.int32)
        for k, r in enumerate(orders(rng, reads) if n else [reads]):
            for flags in (gtx.READS_SORTED, 0):
                got, info = engine.count(r, None, flags)
                np.testing.assert_array_equal(got, want_c, err_msg="count seed=%d n=%d m=%d order=%d flags=%d" % (seed, n, m, k, flags))
                assert info["n_no_class"] == int((reads[:, 0] >= n_classes).sum())
            cov, _ = engine.coverage(r)
            np.testing.assert_array_equal(cov, want_v, err_msg="coverage seed=%d n=%d m=%d order=%d" % (seed, n, m, k))
        if n:
            sel = reads[:, 0] < n_classes
            got, _ = engine.count(reads, w, gtx.READS_SORTED)
            np.testing.assert_array_equal(got, orc.count(refs, reads[sel], w[sel], algo=orc.BIN_INDEX))
            lens = np.full(n_classes, span + 60, dtype=np.int32)
            step = int(rng.choice([1, 7, 25, 1000]))
            size = step * int(rng.choice([1, 2, 20]))
            for prep in ("1", "c"):
                win, _ = engine.scan(reads, lens, step, size, prep)
                want, _ = orc.scan(reads[sel], lens, step, size, prep, algo=0)
                np.testing.assert_array_equal(win, want, err_msg="scan seed=%d step=%d size=%d prep=%s" % (seed, step, size, prep))


def test_fuzz_sorted_semantics_with_zero_length(engine):
    """sorted-merge rules (GTX_ZERO_LENGTH_OK / GTX_REFS_KEEP_ZERO_LENGTH): zero-length reads and regions take part;
    the host-side correction for coinciding zero-length pairs is NOT part of the C ABI, so such pairs are avoided here"""
    rng = np.random.default_rng(77)
    for _ in range(40):
        m, n = int(rng.integers(1, 400)), int(rng.integers(1, 3000))
        rs = np.sort(rng.integers(1, 5000, size=m))
        rl = rng.choice([-1, 0, 3, 40], size=m)                                      # -1 -> zero-length region (start = end+1)
        refs = np.stack([np.zeros(m, dtype=np.int64), rs, rs + rl], axis=1).astype(np.int32)
        qs = np.sort(rng.integers(1, 5000, size=n))
        ql = rng.choice([0, 2, 49], size=n)
        reads = np.stack([np.zeros(n, dtype=np.int64), qs, qs + ql], axis=1).astype(np.int32)
        engine.set_refs(refs, 1, gtx.REFS_KEEP_ZERO_LENGTH)
        got, _ = engine.count(reads, None, gtx.READS_SORTED | gtx.ZERO_LENGTH_OK)
        want = orc.count(refs, reads, algo=orc.SORTED_MERGE)
        np.testing.assert_array_equal(got, want)
