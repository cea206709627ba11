"""Multi-interval (BED12) regions in `count` without -gaps, through the C ABI (gtx_set_ref_blocks, gtx_count_add_regions,
include/gtx.h; kernels in csrc/gtx_pairs.hip): a query counts once for an index region when their envelopes overlap and some
interval of the one overlaps some interval of the other (genomic_intervals.cpp:1167-1172, :5226-5232, :5304-5317).
Expected values: the oracle's CLI (oracle/gtx_oracle.c) on BED12 files written from the same arrays."""
import os
import subprocess

import numpy as np
import pytest

import gtx
from oracle import orc

pytestmark = pytest.mark.gpu

NAMES = ["chr1", "chr2", "chr3"]


def make_set(rng, n, span, multi_frac, exon, intron, max_blocks, single_len, wide=0, n_classes=3):
    """n regions sorted by (class, start): (env triples [n,3], first [n+1], blocks [*,2]); 1-based inclusive coordinates."""
    c = np.sort(rng.integers(0, n_classes, size=n))
    s = rng.integers(1, span, size=n)
    o = np.lexsort((s, c)); c, s = c[o], s[o]
    env = np.zeros((n, 3), dtype=np.int32); first = np.zeros(n + 1, dtype=np.int64); blocks = []
    wide_at = set(rng.choice(n, wide, replace=False).tolist()) if wide else set()
    for i in range(n):
        if i in wide_at:
            iv = [(int(s[i]), int(s[i]) + int(rng.integers(span // 2, span)))]                 # a region that spans most of the class
        elif rng.random() >= multi_frac:
            iv = [(int(s[i]), int(s[i]) + int(rng.integers(0, single_len)))]
        else:
            at = int(s[i]); iv = []
            for _ in range(int(rng.integers(2, max_blocks + 1))):
                sz = int(rng.integers(exon[0], exon[1])); iv.append((at, at + sz - 1)); at += sz + int(rng.integers(intron[0], intron[1]))
        env[i] = (c[i], iv[0][0], iv[-1][1]); blocks += iv; first[i + 1] = len(blocks)
    return env, first, np.array(blocks, dtype=np.int32).reshape(-1, 2)


def write_bed(path, env, first, blocks, labels=None, weights=None):
    with open(path, "w") as f:
        for i in range(len(env)):
            b = blocks[first[i]:first[i + 1]]
            name = labels[i] if labels is not None else (str(int(weights[i])) if weights is not None else "q%d" % i)
            cols = [NAMES[env[i, 0]], str(env[i, 1] - 1), str(env[i, 2]), name, "0", "+"]
            if len(b) > 1:
                cols += [str(env[i, 1] - 1), str(env[i, 2]), "0", str(len(b)), ",".join(str(int(x[1] - x[0] + 1)) for x in b) + ",",
                         ",".join(str(int(x[0] - env[i, 1])) for x in b) + ","]
            f.write("\t".join(cols) + "\n")


def oracle_counts(tmp, refs, reads, extra=(), weights=None):
    write_bed(tmp / "refs.bed", *refs, labels=["r%d" % i for i in range(len(refs[0]))])
    write_bed(tmp / "reads.bed", *reads, weights=weights)
    r = subprocess.run([orc.CLI, "count", "-i"] + list(extra) + ["refs.bed", "reads.bed"], capture_output=True, cwd=tmp)
    assert r.returncode == 0, r.stderr.decode()
    return np.array([int(l.split("\t")[1]) for l in r.stdout.decode().splitlines()], dtype=np.uint64)


def split(reads):
    """single-interval reads as triples; the others as (env, first, blocks) lists"""
    env, first, blocks = reads
    cnt = np.diff(first)
    one = cnt == 1
    m_env = env[~one]; m_cnt = cnt[~one]
    m_first = np.concatenate(([0], np.cumsum(m_cnt)))
    keep = np.repeat(~one, cnt)
    return env[one], (m_env, m_first, blocks[keep])


@pytest.fixture(scope="module")
def sets():
    rng = np.random.default_rng(101)
    genes = make_set(rng, 6000, 1_500_000, 0.7, (50, 300), (100, 20000), 12, 5000, wide=4)
    reads = make_set(rng, 120_000, 1_500_000, 0.3, (10, 80), (50, 5000), 3, 120)
    peaks = make_set(rng, 5000, 1_500_000, 0.0, None, None, 0, 2000, wide=3)
    return genes, reads, peaks


def test_both_sides_multi_interval(engine, sets, tmp_path):
    genes, reads, _ = sets
    want = oracle_counts(tmp_path, genes, reads)
    gaps = oracle_counts(tmp_path, genes, reads, ["-gaps"])
    assert (want != gaps).sum() > 100 and want.sum() < gaps.sum()                  # the two rules differ on this input
    single, multi = split(reads)
    engine.set_refs(genes[0], 3)
    engine.set_ref_blocks(genes[1], genes[2])
    for flags in (gtx.READS_SORTED, 0):
        hits, _ = engine.count_stream([(single[:50000], None), (single[50000:], None)], flags, regions=[(multi[0], None, multi[1], multi[2])])
        np.testing.assert_array_equal(hits, want)
    # the multi-interval reads in two calls, and before the plain ones
    k = len(multi[0]) // 2; f = multi[1]
    parts = [(multi[0][:k], None, f[:k + 1], multi[2][:f[k]]), (multi[0][k:], None, f[k:] - f[k], multi[2][f[k]:])]
    engine.lib.gtx_count_begin(engine.ctx)
    for env, w, first, blocks in parts:
        env = np.ascontiguousarray(env); first = np.ascontiguousarray(first, dtype=np.int64); blocks = np.ascontiguousarray(blocks)
        assert engine.lib.gtx_count_add_regions(engine.ctx, env.ctypes.data, None, first.ctypes.data, blocks.ctypes.data, len(env)) == 0
    assert engine.lib.gtx_count_add(engine.ctx, np.ascontiguousarray(single).ctypes.data, None, len(single), gtx.READS_SORTED) == 0
    out = np.zeros(len(want), dtype=np.uint64)
    assert engine.lib.gtx_count_end(engine.ctx, out.ctypes.data, None) == 0
    np.testing.assert_array_equal(out, want)
    # envelopes only again: the -gaps rule for the plain reads (the multi-interval ones still need their lists)
    engine.set_ref_blocks(None)
    hits, _ = engine.count_stream([(single, None)])
    np.testing.assert_array_equal(hits, oracle_counts(tmp_path, genes, (single, np.arange(len(single) + 1), single[:, 1:3]), ["-gaps"]))


def test_plain_reads_resident_in_hbm(engine, sets, tmp_path):
    """gtx_count_device with multi-interval index regions declared: the reads in a gap of a region are taken off on the device."""
    import torch
    genes, reads, _ = sets
    single, _ = split(reads)
    want = oracle_counts(tmp_path, genes, (single, np.arange(len(single) + 1), single[:, 1:3]))
    engine.set_refs(genes[0], 3)
    engine.set_ref_blocks(genes[1], genes[2])
    d = torch.from_numpy(np.ascontiguousarray(single)).cuda()
    out = torch.zeros(len(want), dtype=torch.int64, device="cuda")
    rng = np.random.default_rng(3)
    w = rng.integers(0, 6, size=len(single)).astype(np.int32)
    for flags in (gtx.READS_SORTED, 0):
        engine.count_device(d.data_ptr(), len(single), out.data_ptr(), None, flags)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64), want)
    dw = torch.from_numpy(w).cuda()
    engine.count_device(d.data_ptr(), len(single), out.data_ptr(), dw.data_ptr(), gtx.READS_SORTED)
    torch.cuda.synchronize()
    wantw = oracle_counts(tmp_path, genes, (single, np.arange(len(single) + 1), single[:, 1:3]), ["--max-label-value", "100"], weights=w)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64), wantw)
    engine.set_ref_blocks(None)


def test_spliced_reads_single_interval_regions_weighted(engine, sets, tmp_path):
    _, reads, peaks = sets
    rng = np.random.default_rng(4)
    w = rng.integers(0, 6, size=len(reads[0])).astype(np.int32)
    want = oracle_counts(tmp_path, peaks, reads, ["--max-label-value", "100"], weights=w)
    cnt = np.diff(reads[1]); one = cnt == 1
    single, multi = split(reads)
    engine.set_refs(peaks[0], 3)
    hits, _ = engine.count_stream([(single, w[one])], regions=[(multi[0], w[~one], multi[1], multi[2])])
    np.testing.assert_array_equal(hits, want)


def test_three_members_on_one_device(sets, tmp_path):
    genes, reads, _ = sets
    want = oracle_counts(tmp_path, genes, reads)
    single, multi = split(reads)
    os.environ["GTX_GROUP_REHEARSE"] = "1"
    try:
        g = gtx.Group([0, 0, 0])
    finally:
        del os.environ["GTX_GROUP_REHEARSE"]
    try:
        g.set_refs(genes[0], 3)
        g.set_ref_blocks(genes[1], genes[2])
        k = len(multi[0]) // 3; f = multi[1]
        parts = [(multi[0][a:b], None, f[a:b + 1] - f[a], multi[2][f[a]:f[b]]) for a, b in ((0, k), (k, 2 * k), (2 * k, len(multi[0])))]
        hits, _ = g.count([(single[:70000], None), (single[70000:], None)], regions=parts)
        np.testing.assert_array_equal(hits, want)
        assert g.member_reads().min() > 0
        g.set_ref_blocks(None)                                                      # back to the members' own shares of the finalize step
        hits, _ = g.count([(single, None)])
        np.testing.assert_array_equal(hits, oracle_counts(tmp_path, genes, (single, np.arange(len(single) + 1), single[:, 1:3]), ["-gaps"]))
    finally:
        g.close()


def test_argument_checks(engine, sets, tmp_path):
    genes, _, _ = sets
    engine.set_refs(genes[0], 3)
    first = genes[1]
    k = int(np.argmax(np.diff(first) > 2))
    bad = genes[2].copy(); bad[first[k] + 1, 0] = bad[first[k], 0] - 5                                          # a start that goes back
    with pytest.raises(gtx.GtxError):
        engine.set_ref_blocks(first, bad)
    bad = genes[2].copy(); bad[first[k], 0] += 1                                                                # not the envelope
    with pytest.raises(gtx.GtxError):
        engine.set_ref_blocks(first, bad)
    env = np.array([[0, 1000, 90000]], dtype=np.int32); f = np.array([0, 2], dtype=np.int64); b = np.array([[1000, 1200], [60000, 90000]], dtype=np.int32)
    assert engine.lib.gtx_count_add_regions(engine.ctx, env.ctypes.data, None, f.ctypes.data, b.ctypes.data, 1) != 0    # no open call
    # multi-interval queries against envelopes that were never declared multi-interval: every region is its one interval
    flat = (genes[0], np.arange(len(genes[0]) + 1), genes[0][:, 1:3])
    hits, _ = engine.count_stream([], regions=[(env, None, f, b)])
    np.testing.assert_array_equal(hits, oracle_counts(tmp_path, flat, (env, f, b)))
    assert 0 < int(hits.sum()) < int(oracle_counts(tmp_path, flat, (env, f, b), ["-gaps"]).sum())


def test_edges(engine, tmp_path):
    """No regions at all, classes nobody knows, a region list without any multi-interval region, the one-shot call."""
    env = np.array([[0, 10, 50], [7, 10, 50], [-1, 10, 50]], dtype=np.int32); f = np.array([0, 2, 4, 6], dtype=np.int64)
    b = np.array([[10, 20], [30, 50]] * 3, dtype=np.int32)
    engine.set_refs(np.zeros((0, 3), dtype=np.int32), 3)
    engine.set_ref_blocks(np.zeros(1, dtype=np.int64), np.zeros((0, 2), dtype=np.int32))
    hits, _ = engine.count_stream([(np.array([[0, 1, 5]], dtype=np.int32), None)], regions=[(env, None, f, b)])
    assert len(hits) == 0
    refs = np.array([[0, 15, 25], [0, 21, 29], [1, 1, 100], [0, 45, 60]], dtype=np.int32)
    engine.set_refs(refs, 3)
    engine.set_ref_blocks(np.arange(5), refs[:, 1:3])                      # every region its one interval: nothing to correct
    hits, _ = engine.count_stream([], regions=[(env, None, f, b)])          # classes 7 and -1 match nothing
    assert hits.tolist() == [1, 0, 0, 1]                                   # [10,20] meets [15,25]; [21,29] lies in the gap; [30,50] meets [45,60]
    blocks = np.array([[15, 17], [24, 25], [21, 22], [28, 29], [1, 100], [45, 60]], dtype=np.int32)
    engine.set_ref_blocks(np.array([0, 2, 4, 5, 6]), blocks)
    hits, _ = engine.count(np.array([[0, 18, 23], [0, 23, 23], [0, 16, 16], [1, 5, 5]], dtype=np.int32))   # gtx_count, one shot
    assert hits.tolist() == [1, 1, 1, 0]                                   # region 0: only [16,16]; region 1: only [18,23] ([21,22]); [23,23] lies in both gaps
    engine.set_ref_blocks(None)


def test_large_batch_properties_and_one_class_against_the_oracle(engine, tmp_path):
    """20 M plain reads resident in HBM x 200 k regions of which a third are multi-interval: regions with one interval keep their
    envelope count, the others can only lose reads; one class (the smallest) against the oracle CLI read by read."""
    import torch
    from gtx import synth
    from bench import make_reads_on_device
    rng = np.random.default_rng(9)
    refs = synth.genome_intervals(200_000, 51, 200, 4000)
    multi = rng.random(len(refs)) < 0.33
    first = np.zeros(len(refs) + 1, dtype=np.int64); blocks = []
    for k in range(len(refs)):
        s, e = int(refs[k, 1]), int(refs[k, 2])
        if multi[k] and e - s >= 40:
            cuts = np.sort(rng.choice(np.arange(s + 1, e), size=4, replace=False))
            blocks += [(s, int(cuts[0])), (int(cuts[1]), int(cuts[2])), (int(cuts[3]), e)]
        else:
            multi[k] = False
            blocks.append((s, e))
        first[k + 1] = len(blocks)
    blocks = np.array(blocks, dtype=np.int32)
    dev = torch.device("cuda", 0)
    n = 20_000_000
    reads = make_reads_on_device(n, np.arange(24), 1000, dev)
    out = torch.zeros(len(refs), dtype=torch.int64, device=dev)
    engine.set_refs(refs, 24)
    engine.count_device(reads.data_ptr(), n, out.data_ptr(), None, gtx.READS_SORTED); torch.cuda.synchronize()
    env = out.cpu().numpy().copy()
    engine.set_ref_blocks(first, blocks)
    engine.count_device(reads.data_ptr(), n, out.data_ptr(), None, gtx.READS_SORTED); torch.cuda.synchronize()
    got = out.cpu().numpy().copy()
    assert np.array_equal(got[~multi], env[~multi])
    assert np.all(got[multi] <= env[multi]) and int((env - got).sum()) > 10_000
    # one class read by read: chr21 (class id by name order)
    r = reads.cpu().numpy()
    cls = int(np.argmin(np.bincount(r[:, 0], minlength=24)))
    sub = r[r[:, 0] == cls]
    keep = refs[:, 0] == cls
    idx = np.nonzero(keep)[0]
    sref = refs[keep].copy(); sref[:, 0] = 0
    sfirst = np.concatenate(([0], np.cumsum(np.diff(first)[keep])))
    sblocks = np.concatenate([blocks[first[k]:first[k + 1]] for k in idx])
    sreads = sub.copy(); sreads[:, 0] = 0
    want = oracle_counts(tmp_path, (sref, sfirst, sblocks), (sreads, np.arange(len(sreads) + 1), sreads[:, 1:3]))
    np.testing.assert_array_equal(got[keep].view(np.uint64), want)
    engine.set_ref_blocks(None)
