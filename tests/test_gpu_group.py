"""gtx_group_* (include/gtx.h): the counting path on several GPUs of one node -- classes dealt to the members, RCCL reduce
of the result vector.  One GPU is all a test box has, so the members sit on the same device under GTX_GROUP_REHEARSE=1
(routing, per-member streams and the finish are the real code; only ncclReduce is replaced by an add kernel), and a group
of ONE member runs through real RCCL calls (communicator + ncclReduce) under GTX_GROUP_FORCE_RCCL=1."""
import os
import subprocess

import numpy as np
import pytest

import gtx
from gtx import synth
from oracle import orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd", "csrc")


@pytest.fixture(scope="module")
def rehearsal_group():
    os.environ["GTX_GROUP_REHEARSE"] = "1"
    try:
        g = gtx.Group([0, 0, 0])
    finally:
        del os.environ["GTX_GROUP_REHEARSE"]
    yield g
    g.close()


def test_sorted_reads_go_to_their_owners_in_runs(rehearsal_group):
    g = rehearsal_group
    rng = np.random.default_rng(5)
    refs = synth.genome_intervals(30000, 81, 50, 2000)
    reads = synth.genome_intervals(600_000, 82, 50, 51)
    w = rng.integers(0, 5, size=len(reads)).astype(np.int32)
    g.set_refs(refs, synth.n_classes())
    per = np.bincount(reads[:, 0], minlength=24)
    owner = g.assign(per)
    cuts = [0, 200_000, 200_001, 450_000, len(reads)]
    batches = [(reads[a:b], None) for a, b in zip(cuts[:-1], cuts[1:])]
    hits, info = g.count(batches)
    np.testing.assert_array_equal(hits, orc.count(refs, reads, algo=orc.SORTED_MERGE))
    np.testing.assert_array_equal(g.member_reads(), np.bincount(owner[reads[:, 0]], minlength=3))
    assert g.member_reads().min() > 0 and info["n_no_class"] == 0
    hits, _ = g.count([(reads[a:b], w[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    np.testing.assert_array_equal(hits, orc.count(refs, reads, w))
    cov, _ = g.coverage([(reads, w)])
    np.testing.assert_array_equal(cov, orc.coverage(refs, reads, w))


def test_interleaved_reads_are_partitioned(rehearsal_group):
    g = rehearsal_group
    rng = np.random.default_rng(6)
    refs = synth.genome_intervals(20000, 83, 50, 2000)
    reads = synth.genome_intervals(300_000, 84, 50, 51)
    reads = reads[rng.permutation(len(reads))]
    reads[1000, 0] = 200                                          # no such class: counted as such by member 0
    g.set_refs(refs, synth.n_classes())
    hits, info = g.count([(reads, None)], flags=0)
    np.testing.assert_array_equal(hits, orc.count(refs, np.delete(reads, [1000], axis=0), algo=orc.BIN_INDEX))
    assert info["n_no_class"] == 1 and int(g.member_reads().sum()) == len(reads)


def test_default_assignment_and_scan(rehearsal_group):
    g = rehearsal_group
    g2 = None
    os.environ["GTX_GROUP_REHEARSE"] = "1"
    try:
        g2 = gtx.Group([0, 0])                                     # a fresh group: no gtx_group_assign call, assignment by reference span
    finally:
        del os.environ["GTX_GROUP_REHEARSE"]
    refs = synth.genome_intervals(20000, 85, 50, 2000, stranded=True)
    reads = synth.genome_intervals(250_000, 86, 50, 51, stranded=True)
    g2.set_refs(refs, synth.n_classes(True))
    hits, _ = g2.count([(reads, None)])
    np.testing.assert_array_equal(hits, orc.count(refs, reads, algo=orc.BIN_INDEX))
    assert g2.member_reads().min() > 50_000
    plain = synth.genome_intervals(250_000, 87, 50, 51)
    for step, size in ((1000, 1000), (25, 500)):
        win, _ = g2.scan(plain, synth.CHROM_LEN, step, size)
        want, _ = orc.scan(plain, synth.CHROM_LEN, step, size)
        np.testing.assert_array_equal(win, want)
    assert g2.member_reads().min() > 50_000
    win, _ = g2.scan(plain[np.random.default_rng(1).permutation(len(plain))], synth.CHROM_LEN, 1000, 2000)   # owners interleave: partitioned
    want, _ = orc.scan(plain, synth.CHROM_LEN, 1000, 2000)
    np.testing.assert_array_equal(win, want)
    g2.close()


def test_group_of_one_through_rccl():
    """communicator creation and ncclReduce for real (one rank: in place at the root)"""
    os.environ["GTX_GROUP_FORCE_RCCL"] = "1"
    try:
        g = gtx.Group([0])
    finally:
        del os.environ["GTX_GROUP_FORCE_RCCL"]
    refs = synth.genome_intervals(20000, 88, 50, 2000)
    reads = synth.genome_intervals(300_000, 89, 50, 51)
    g.set_refs(refs, synth.n_classes())
    hits, _ = g.count([(reads, None)])
    np.testing.assert_array_equal(hits, orc.count(refs, reads, algo=orc.SORTED_MERGE))
    win, _ = g.scan(reads, synth.CHROM_LEN, 1000, 1000)
    want, _ = orc.scan(reads, synth.CHROM_LEN, 1000, 1000)
    np.testing.assert_array_equal(win, want)
    g.close()


def test_group_errors():
    lib = gtx.load()
    assert not lib.gtx_group_create(0, None)
    assert b"at least one device" in lib.gtx_group_last_error(None)
    ids = np.array([0, 0], dtype=np.int32)
    assert not lib.gtx_group_create(2, ids.ctypes.data)             # the same device twice (outside the rehearsal mode)
    assert b"twice" in lib.gtx_group_last_error(None)
    g = gtx.Group([0])
    with pytest.raises(gtx.GtxError):
        g.count([(np.zeros((1, 3), dtype=np.int32), None)])          # no reference set yet
    g.close()


def _bed(path, tri, names, labels=None):
    with open(path, "w") as f:
        for i, (c, s, e) in enumerate(tri):
            f.write("%s\t%d\t%d\t%s\t0\t+\n" % (names[c], int(s) - 1, int(e), "r%d" % i if labels is None else labels[i]))


def test_cli_ngpu(tmp_path):
    """genomic_overlaps / genomic_scans --ngpu N: same bytes as one GPU (3 members rehearsed on device 0, and one member
    through RCCL)."""
    refs = synth.genome_intervals(3000, 91, 50, 4000)
    reads = synth.genome_intervals(80000, 92, 30, 300)
    lab = np.random.default_rng(3).integers(0, 5, size=len(reads))
    _bed(tmp_path / "refs.bed", refs, synth.CHROM_NAMES)
    _bed(tmp_path / "reads.bed", reads, synth.CHROM_NAMES, lab)
    with open(tmp_path / "genome.bed", "w") as f:
        for n, ln in zip(synth.CHROM_NAMES, synth.CHROM_LEN):
            f.write("%s\t0\t%d\n" % (n, ln))

    def run(tool, args, env_extra):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([os.path.join(BIN, tool)] + args, capture_output=True, cwd=tmp_path, env=env)
        assert r.returncode == 0, r.stderr.decode()
        return r.stdout
    for args in (["count", "-S", "-i", "refs.bed", "reads.bed"], ["count", "-i", "--max-label-value", "4", "refs.bed", "reads.bed"],
                 ["coverage", "-S", "refs.bed", "reads.bed"]):
        one = run("genomic_overlaps", args, {})
        assert one == orc.cli([str(tmp_path / a) if a.endswith('.bed') else a for a in args])[1].encode()
        assert run("genomic_overlaps", [args[0], "--ngpu", "3"] + args[1:], {"GTX_GROUP_REHEARSE": "1"}) == one
        assert run("genomic_overlaps", [args[0], "--ngpu", "1"] + args[1:], {"GTX_GROUP_FORCE_RCCL": "1"}) == one
        assert run("genomic_overlaps", args, {"GTX_GROUP_REHEARSE": "1", "GTX_NGPU": "2"}) == one
    sargs = ["counts", "-i", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "2", "reads.bed"]
    one = run("genomic_scans", sargs, {})
    assert len(one) > 1000
    assert run("genomic_scans", [sargs[0], "--ngpu", "3"] + sargs[1:], {"GTX_GROUP_REHEARSE": "1"}) == one
