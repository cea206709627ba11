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
