"""Parity of the HIP scan path (gtx_scan / gtx_scan_device) against the CPU oracle's two scanners -- bit-exact."""
import numpy as np
import pytest

import gtx
from gtx import synth
from oracle import orc

pytestmark = pytest.mark.gpu

LENS = synth.CHROM_LEN // 20          # scaled-down genome keeps the micro-window arrays small


def reads_scaled(n, seed, stranded=False):
    r = synth.genome_intervals(n, seed, 50, 51, stranded=stranded)
    r[:, 1] = r[:, 1] // 20 + 1
    r[:, 2] = r[:, 1] + 49
    return r[np.lexsort((r[:, 1], r[:, 0]))]


@pytest.mark.parametrize("step,size", [(1000, 1000), (25, 500), (100, 300), (7, 7 * 13)])
@pytest.mark.parametrize("prep", ["1", "c"])
def test_scan_matches_unsorted_scanner(engine, step, size, prep):
    reads = reads_scaled(300000, 41)
    got, off = engine.scan(reads, LENS, step, size, prep)
    want, woff = orc.scan(reads, LENS, step, size, prep, algo=0)
    np.testing.assert_array_equal(off, woff)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("step,size", [(1000, 1000), (25, 500)])
def test_scan_matches_sorted_scanner(engine, step, size):
    reads = reads_scaled(200000, 42)
    got, _ = engine.scan(reads, LENS, step, size, "1", flags=gtx.ZERO_LENGTH_OK)
    want, _ = orc.scan(reads, LENS, step, size, "1", algo=1)
    np.testing.assert_array_equal(got, want)


def test_scan_unsorted_input_and_weights(engine):
    rng = np.random.default_rng(43)
    reads = reads_scaled(200000, 43)
    reads = reads[rng.permutation(len(reads))]
    w = rng.integers(0, 5, size=len(reads)).astype(np.int32)
    got, _ = engine.scan(reads, LENS, 200, 1000, "1", weights=w)
    want, _ = orc.scan(reads, LENS, 200, 1000, "1", weights=w, algo=0)
    np.testing.assert_array_equal(got, want)


def test_scan_sorted_weighted_uses_the_lds_tile(engine):
    """Weighted sorted reads: the 64-bit per-wave LDS tile follows the reads; large and negative label values wrap in 64 bits
    like the reference's unsigned long; fine and coarse window steps; reads of several classes."""
    rng = np.random.default_rng(47)
    reads = reads_scaled(300000, 47)
    reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    for w in (rng.integers(0, 5, size=len(reads)), rng.integers(-(1 << 31), (1 << 31) - 1, size=len(reads))):
        w = w.astype(np.int32)
        for step, size, prep in ((200, 1000, "1"), (25, 500, "c"), (5000, 5000, "1")):
            got, _ = engine.scan(reads, LENS, step, size, prep, weights=w)
            want, _ = orc.scan(reads, LENS, step, size, prep, weights=w, algo=0)
            np.testing.assert_array_equal(got, want)


def test_scan_edges(engine):
    # reads beyond the last whole micro-window, class without windows, unknown class, invalid reads
    lens = np.array([10000, 2500, 700], dtype=np.int32)
    reads = np.array([[0, 1, 50], [0, 1000, 1049], [0, 1001, 1050], [0, 2501, 2550], [0, 10000, 10049], [0, 10001, 10050],
                      [1, 101, 150], [1, 2101, 2150], [2, 11, 60], [3, 11, 60], [0, 500, 400], [0, -20, 0], [0, -5, 30]], dtype=np.int32)
    got, off = engine.scan(reads, lens, 1000, 1000, "1")
    want, _ = orc.scan(reads, lens, 1000, 1000, "1", algo=0)
    np.testing.assert_array_equal(got, want)
    assert got[:3].tolist() == [2, 1, 1] and got[9] == 1 and got[off[1]] == 1
    got2, _ = engine.scan(reads, lens, 1000, 2000, "1")
    want2, _ = orc.scan(reads, lens, 1000, 2000, "1", algo=0)
    np.testing.assert_array_equal(got2, want2)


def test_scan_empty(engine):
    got, _ = engine.scan(np.zeros((0, 3), dtype=np.int32), LENS, 1000, 1000)
    assert int(got.sum()) == 0


def test_scan_device_entry(engine):
    torch = pytest.importorskip("torch")
    reads = reads_scaled(300000, 44)
    want, off = engine.scan(reads, LENS, 1000, 1000)
    d_reads = torch.from_numpy(reads).cuda()
    d_out = torch.zeros(len(want), dtype=torch.int64, device="cuda")
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    engine.scan_device(d_reads.data_ptr(), len(reads), LENS, 1000, 1000, d_out.data_ptr())
    engine.sync()
    engine.set_stream(0)
    np.testing.assert_array_equal(d_out.cpu().numpy().view(np.uint64), want)


def test_scan_total_is_preserved_full_size_property(engine):
    """Size-independent property at a large size: with window == step every in-range read lands in exactly one window."""
    reads = synth.genome_intervals(5_000_000, 45, 50, 51)
    got, _ = engine.scan(reads, synth.CHROM_LEN, 1000, 1000)
    inside = 0
    for c, ln in enumerate(synth.CHROM_LEN):
        s = reads[reads[:, 0] == c, 1].astype(np.int64)
        inside += int(((s - 1) // 1000 < ln // 1000).sum())
    assert int(got.sum()) == inside


# ---- owner-computes pass for reads in (class, start) order (gtx_scanown.hip), selected by GTX_READS_SORTED ----------------------
SORTED = 1   # gtx.READS_SORTED


@pytest.fixture(autouse=True)
def owner_pass_on_any_geometry(monkeypatch):
    # the library takes the owner pass only where it pays (few reads per micro-window); the tests want it everywhere the hint is given
    monkeypatch.setenv("GTX_SCAN_OWN_ALWAYS", "1")


@pytest.mark.parametrize("step,size", [(1000, 1000), (25, 500), (100, 300), (1, 7), (5000, 20000), (25, 25 * 1500)])
def test_sorted_hint_takes_the_owner_pass(engine, step, size):
    """Sorted reads with the hint: every block owns a range of windows.  Same numbers as both restated scanners, with the
    unsorted scanner's rules and with the sorted scanner's (GTX_ZERO_LENGTH_OK), weighted and not, through the host entry
    (reads made resident, one launch) -- including window steps of 1 bp and windows of 1500 steps."""
    rng = np.random.default_rng(step + size)
    small = (np.asarray(synth.CHROM_LEN) // 200).astype(np.int64)          # a genome of ~15 Mb keeps the 1-bp case small
    reads = synth.genome_intervals(200_000, 5 + step, 30, 120)
    reads[:, 1:] = reads[:, 1:] // 200 + 1
    reads[:, 2] = reads[:, 1] + rng.integers(0, 90, size=len(reads))
    reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    reads[7, 2] = reads[7, 1] - 1                                          # a zero-length read
    reads[9, 1] = -4                                                       # start below 1 (first read region of chr1 anyway)
    reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    w = rng.integers(-1, 5, size=len(reads)).astype(np.int32)
    for ww in (None, w):
        for fl, algo in ((SORTED, 0), (SORTED | gtx.ZERO_LENGTH_OK, 1)):
            got, _ = engine.scan(reads, small, step, size, "1", ww, fl)
            want, _ = orc.scan(reads, small, step, size, "1", ww, algo=algo)
            np.testing.assert_array_equal(got, want)


def test_sorted_hint_on_unsorted_reads_falls_back(engine):
    """The hint is checked on the device: reads that are NOT in order (shuffled; one swap far into the stream; two sorted halves;
    classes interleaved) give the general kernels' numbers -- exactly -- through the conditional fallback."""
    rng = np.random.default_rng(77)
    reads = synth.genome_intervals(400_000, 78, 50, 51)
    want, _ = orc.scan(reads, synth.CHROM_LEN, 25, 500)
    variants = [reads[rng.permutation(len(reads))]]
    one = reads.copy(); one[[300_000, 300_001]] = one[[300_001, 300_000]]
    if one[300_000, 1] == one[300_001, 1]:
        one[300_001, 1] -= 1
    variants.append(one)
    variants.append(np.concatenate([reads[200_000:], reads[:200_000]]))
    variants.append(reads[np.lexsort((reads[:, 0], reads[:, 1]))])         # by start only: classes interleave
    for v in variants:
        got, _ = engine.scan(v, synth.CHROM_LEN, 25, 500, "1", None, SORTED)
        np.testing.assert_array_equal(got, want)
    got, _ = engine.scan(reads, synth.CHROM_LEN, 25, 500, "1", None, SORTED)     # and a sorted call right after a fallback
    np.testing.assert_array_equal(got, want)


def test_owner_pass_class_edge_cases(engine):
    """Classes without windows (shorter than one window), without reads, reads beyond a class's last micro-window, unknown and
    negative class ids around the known ones, a single read, no reads at all."""
    lens = np.array([5000, 300, 0, 100_000, 999, 40_000], dtype=np.int32)
    rng = np.random.default_rng(79)
    parts = []
    for c, ln in ((-2, 1000), (0, 5000), (1, 300), (3, 100_000), (3, 140_000), (4, 999), (5, 40_000), (9, 1000)):
        k = 3000 if ln > 1000 else 50
        s = np.sort(rng.integers(1, max(ln, 2), size=k))
        parts.append(np.stack([np.full(k, c), s, s + rng.integers(0, 60, size=k)], axis=1))
    reads = np.concatenate(parts).astype(np.int32)
    reads = reads[np.lexsort((reads[:, 1], reads[:, 0]))]
    known = (reads[:, 0] >= 0) & (reads[:, 0] < len(lens))
    for step, size in ((25, 500), (1000, 1000), (100, 1000)):
        got, _ = engine.scan(reads, lens, step, size, "1", None, SORTED)
        want, _ = orc.scan(reads[known], lens, step, size)
        np.testing.assert_array_equal(got, want)
    for sub in (reads[:0], reads[known][:1]):
        got, _ = engine.scan(sub, lens, 25, 500, "1", None, SORTED)
        want, _ = orc.scan(sub, lens, 25, 500)
        np.testing.assert_array_equal(got, want)


def test_owner_pass_device_entry_at_scale(engine):
    """5 M reads resident in HBM, default window geometry, device entry: owner pass == general kernels (no hint) == oracle sample."""
    torch = pytest.importorskip("torch")
    reads = synth.genome_intervals(5_000_000, 81, 50, 51)
    d = torch.from_numpy(reads).cuda()
    off, tot = gtx.scan_layout(synth.CHROM_LEN, 25, 500)
    a = torch.zeros(tot, dtype=torch.int64, device="cuda"); b = torch.zeros_like(a)
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    engine.scan_device(d.data_ptr(), len(reads), synth.CHROM_LEN, 25, 500, a.data_ptr(), flags=SORTED)
    engine.scan_device(d.data_ptr(), len(reads), synth.CHROM_LEN, 25, 500, b.data_ptr(), flags=0)
    engine.sync(); engine.set_stream(0)
    assert torch.equal(a, b)
    assert int(a.sum().item()) > 20 * 4_000_000


def test_owner_pass_is_chosen_by_geometry(engine, monkeypatch):
    """Without the test override: coarse steps (many reads per micro-window) keep the general kernels, fine steps take the owner
    pass -- either way the numbers are the oracle's."""
    monkeypatch.delenv("GTX_SCAN_OWN_ALWAYS")
    reads = reads_scaled(300000, 49)
    for step, size in ((5000, 5000), (10, 200)):
        got, _ = engine.scan(reads, LENS, step, size, "1", None, SORTED)
        want, _ = orc.scan(reads, LENS, step, size, "1")
        np.testing.assert_array_equal(got, want)


def test_config4_100m_reads_both_geometries(engine, monkeypatch):
    """BASELINE config 4 at full size on one GPU: 100 M reads resident in HBM, `-w 1000 -d 1000` and the default `-w 500 -d 25`.
    The CPU oracle does not run at this size, so: the owner-computes pass and the general kernels (two independent code paths) give the
    same vector, every in-range read lands in exactly one window when window == step, every window of the fine geometry is the sum
    of its micro-windows (checked through the 1-step geometry on one chromosome), and one chromosome's windows equal the oracle's
    on that chromosome's ~1.5 M reads."""
    torch = pytest.importorskip("torch")
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("GTX_SCAN_OWN_ALWAYS", raising=False)                 # the library's own choice of pass
    n = 100_000_000
    dev = torch.device("cuda", 0)
    reads = bench.make_reads_on_device(n, np.arange(24), 1000, dev)
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    c = int(np.argmin(synth.CHROM_LEN))
    sub = reads[reads[:, 0] == c].cpu().numpy()
    for step, size in ((1000, 1000), (25, 500)):
        off, tot = gtx.scan_layout(synth.CHROM_LEN, step, size)
        a = torch.zeros(tot, dtype=torch.int64, device=dev); b = torch.zeros_like(a)
        engine.scan_device(reads.data_ptr(), n, synth.CHROM_LEN, step, size, a.data_ptr(), flags=SORTED)
        engine.scan_device(reads.data_ptr(), n, synth.CHROM_LEN, step, size, b.data_ptr(), flags=0)
        engine.sync()
        assert torch.equal(a, b)
        if step == size:                                                         # every read that starts inside a whole window lands in exactly one
            lens = torch.from_numpy(np.asarray(synth.CHROM_LEN, dtype=np.int64)).to(dev)
            inside = ((reads[:, 1].long() - 1) // step < lens[reads[:, 0].long()] // step).sum()
            assert int(a.sum().item()) == int(inside.item())
        want, woff = orc.scan(sub, synth.CHROM_LEN, step, size)
        nw = gtx.load().gtx_scan_n_windows(int(synth.CHROM_LEN[c]), step, size)
        np.testing.assert_array_equal(a[off[c]:off[c] + nw].cpu().numpy().view(np.uint64), want[woff[c]:woff[c] + nw])
    engine.set_stream(0)


def _bed_text(reads, names, strands=None, labels=None):
    """packed triples (1-based inclusive) -> BED6 lines"""
    out = []
    for i, (c, s, e) in enumerate(reads):
        out.append("%s\t%d\t%d\t%s\t0\t%s\n" % (names[c % len(names)], s - 1, e, "r" if labels is None else str(labels[i]),
                                              "+" if strands is None else strands[i]))
    return "".join(out).encode()


@pytest.mark.gpu
@pytest.mark.parametrize("step,size,prep", [(1000, 1000, "1"), (25, 500, "1"), (1000, 2000, "c")])
def test_scan_fed_as_a_stream(step, size, prep):
    """gtx_scan_begin .. gtx_scan_end (what genomic_scans uses on one GPU): packed host batches -- one of them shuffled -- and blocks of
    BED text tokenised on the device must add up to the all-at-once scan and to the oracle's; a text block with a line the device
    does not take (a space-separated one) comes back whole and leaves nothing behind."""
    rng = np.random.default_rng(17)
    e = gtx.Engine(0)
    a = synth.genome_intervals(300_000, 31, 50, 51); b = synth.genome_intervals(200_000, 32, 50, 300)
    c = synth.genome_intervals(40_000, 33, 50, 51); d = synth.genome_intervals(30_000, 34, 50, 51)
    b = b[rng.permutation(len(b))]
    c[5, 1], c[5, 2] = 700, 600                                             # start > stop: the unsorted scanner skips it
    names = synth.CHROM_NAMES
    rules = gtx.TextRules.make(names)
    odd = _bed_text(d, names).replace(b"\t", b" ", 5)                       # its first line is space-separated: not the device's case
    pieces = [(a, None, 0), (b, None, gtx.READS_UNSORTED), (_bed_text(c, names), rules, 0), (odd, rules, 0)]
    win, off, labels, verdicts = e.scan_stream(pieces, synth.CHROM_LEN, step, size, preprocess=prep)
    assert verdicts == [0, 1] and labels == len(c)
    allr = np.concatenate([a, b, c])
    want, _ = orc.scan(allr, synth.CHROM_LEN, step, size, preprocess=prep)
    np.testing.assert_array_equal(win, want)
    once, _ = e.scan(allr, synth.CHROM_LEN, step, size, preprocess=prep)
    np.testing.assert_array_equal(once, want)
    # label weights: min(max, atol(column 4)) summed over every line the device takes, windows weighted
    wl = rng.integers(0, 9, size=len(c))
    rules_w = gtx.TextRules.make(names, max_label_value=5)
    wa = rng.integers(0, 6, size=len(a)).astype(np.int32)
    win, off, labels, verdicts = e.scan_stream([(a, wa, 0), (_bed_text(c, names, labels=wl), rules_w, 0)], synth.CHROM_LEN, step, size, preprocess=prep, weighted=True)
    assert verdicts == [0] and labels == int(np.minimum(wl, 5).sum())
    want, _ = orc.scan(np.concatenate([a, c]), synth.CHROM_LEN, step, size, preprocess=prep, weights=np.concatenate([wa, np.minimum(wl, 5).astype(np.int32)]))
    np.testing.assert_array_equal(win, want)
    e.close()
