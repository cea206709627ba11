"""The product CLIs (genomic_overlaps count|rpkm, genomic_scans counts -- C++ host over libgtx.so) against
(1) the known-answer manifest and (2) the oracle CLI on seeded BED text, byte for byte."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from gtx import synth
from oracle import orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd", "csrc")
TOOLS = {"overlaps": os.path.join(BIN, "genomic_overlaps"), "scans": os.path.join(BIN, "genomic_scans")}
CASES = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]


def product(tool, args, stdin=None, cwd=None):
    r = subprocess.run([TOOLS[tool]] + list(args), input=stdin, capture_output=True, cwd=cwd)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def oracle(args, stdin=None, cwd=None):
    r = subprocess.run([orc.CLI] + list(args), input=stdin, capture_output=True, cwd=cwd)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_cli_matches_known_answers(case):
    stdin = open(os.path.join(GOLD, case["stdin_file"]), "rb").read() if "stdin_file" in case else None
    rc, out, err = product(case["tool"], case["args"], stdin, cwd=GOLD)
    assert rc == case["rc"], err
    assert out == case["stdout"]
    assert case.get("stderr_contains", "") in err


def write_bed(path, tri, names, labels=None, strands=None, sep="\t", compress=False):
    """1-based inclusive triples -> BED6 text (0-based start)."""
    lines = []
    for i, (c, s, e) in enumerate(tri):
        lab = "r%d" % i if labels is None else str(labels[i])
        st = "+" if strands is None else strands[i]
        lines.append(sep.join([names[c], str(int(s) - 1), str(int(e)), lab, "0", st]))
    data = ("\n".join(lines) + "\n").encode()
    if compress:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


@pytest.fixture(scope="module")
def beds(tmp_path_factory):
    d = tmp_path_factory.mktemp("beds")
    rng = np.random.default_rng(31)
    names = synth.CHROM_NAMES
    # reference regions: file order = sorted by (chrom, start) so that -S accepts them
    refs = synth.genome_intervals(3000, 31, 50, 4000)
    rstr = rng.choice(["+", "-"], size=len(refs))
    write_bed(d / "refs.bed", refs, names, ["g%d" % i for i in range(len(refs))], rstr)
    reads = synth.genome_intervals(60000, 32, 30, 300)
    qstr = rng.choice(["+", "-"], size=len(reads))
    labels = rng.integers(0, 6, size=len(reads))
    write_bed(d / "reads_pos.bed", reads, names, labels, qstr)                       # sorted by chrom,start (sortbed -i)
    order = np.lexsort((reads[:, 1], qstr == "-", reads[:, 0]))
    write_bed(d / "reads_strand.bed", reads[order], names, labels[order], qstr[order])   # chrom,strand,start (sortbed)
    rorder = np.lexsort((refs[:, 1], rstr == "-", refs[:, 0]))
    write_bed(d / "refs_strand.bed", refs[rorder], names, ["g%d" % i for i in rorder], rstr[rorder])
    perm = rng.permutation(len(reads))
    write_bed(d / "reads_shuffled.bed.gz", reads[perm], names, labels[perm], qstr[perm], compress=True)
    with open(d / "genome.bed", "w") as f:
        for n, ln in zip(names, synth.CHROM_LEN // 50):
            f.write("%s\t0\t%d\n" % (n, ln))
    small = synth.genome_intervals(40000, 33, 50, 51)
    small[:, 1:] = small[:, 1:] // 50 + 1
    small[:, 2] = small[:, 1] + 49
    small = small[np.lexsort((small[:, 1], small[:, 0]))]
    sstr = rng.choice(["+", "-"], size=len(small))
    write_bed(d / "scan_pos.bed", small, names, None, sstr)
    o2 = np.lexsort((small[:, 1], sstr == "-", small[:, 0]))
    write_bed(d / "scan_strand.bed", small[o2], names, None, sstr[o2])
    # reference regions for `counts -r` in the scaled genome: sorted by (chrom,start), by (chrom,strand,start), shuffled
    sref = synth.genome_intervals(400, 34, 200, 60000)
    sref[:, 1:] = sref[:, 1:] // 50 + 1
    sref = sref[np.lexsort((sref[:, 1], sref[:, 0]))]
    sref = sref[sref[:, 0] != 5]                                                     # one chromosome without reference regions
    rs = rng.choice(["+", "-"], size=len(sref))
    write_bed(d / "scan_refs.bed", sref, names, ["r%d" % i for i in range(len(sref))], rs)
    o3 = np.lexsort((sref[:, 1], rs == "-", sref[:, 0]))
    write_bed(d / "scan_refs_strand.bed", sref[o3], names, ["r%d" % i for i in o3], rs[o3])
    p3 = rng.permutation(len(sref))
    write_bed(d / "scan_refs_shuffled.bed", sref[p3], names, ["r%d" % i for i in p3], rs[p3])
    # peaks: control = background, signal = background + 60 clusters of ~40 reads within 400 bp; both sorted by (chrom,start)
    def reads50(n, seed):
        r = synth.genome_intervals(n, seed, 50, 51)
        r[:, 1:] = r[:, 1:] // 50 + 1
        r[:, 2] = r[:, 1] + 49
        return r
    ctrl = reads50(30000, 35)
    bg = reads50(30000, 36)
    centers = reads50(60, 37)
    cl = np.repeat(centers, 40, axis=0)
    cl[:, 1] += rng.integers(0, 400, size=len(cl)); cl[:, 2] = cl[:, 1] + 49
    sig = np.concatenate([bg, cl])
    for name, arr in (("peaks_signal", sig), ("peaks_control", ctrl)):
        arr = arr[np.lexsort((arr[:, 1], arr[:, 0]))]
        st = rng.choice(["+", "-"], size=len(arr))
        lab = rng.integers(0, 4, size=len(arr))
        write_bed(d / (name + ".bed"), arr, names, lab, st)
        o = np.lexsort((arr[:, 1], st == "-", arr[:, 0]))
        write_bed(d / (name + "_strand.bed"), arr[o], names, lab[o], st[o])
    return d


OVERLAP_RUNS = [
    (["count", "refs.bed", "reads_pos.bed"]),
    (["count", "-i", "refs.bed", "reads_pos.bed"]),
    (["count", "-S", "-i", "refs.bed", "reads_pos.bed"]),
    (["count", "-S", "refs.bed", "reads_pos.bed"]),
    (["count", "-S", "-s", "refs_strand.bed", "reads_strand.bed"]),
    (["count", "-i", "refs.bed", "reads_shuffled.bed.gz"]),
    (["count", "refs.bed", "reads_shuffled.bed.gz"]),
    (["count", "-i", "--max-label-value", "4", "refs.bed", "reads_pos.bed"]),
    (["count", "-S", "-i", "--max-label-value", "3", "-min", "5", "refs.bed", "reads_pos.bed"]),
    (["count", "-i", "-gaps", "-min", "2", "refs.bed", "reads_pos.bed"]),
    (["rpkm", "-i", "refs.bed", "reads_pos.bed"]),
    (["rpkm", "-S", "refs.bed", "reads_pos.bed"]),
    (["count", "-S", "-i", "refs.bed", "reads_shuffled.bed.gz"]),       # not sorted -> the reference's error
    (["coverage", "-i", "refs.bed", "reads_pos.bed"]),
    (["coverage", "refs.bed", "reads_pos.bed"]),
    (["coverage", "-S", "-i", "-min", "200", "refs.bed", "reads_pos.bed"]),
    (["coverage", "-S", "-s", "refs_strand.bed", "reads_strand.bed"]),
    (["coverage", "-i", "--max-label-value", "4", "refs.bed", "reads_shuffled.bed.gz"]),
    (["coverage", "-i", "-gaps", "refs.bed", "reads_pos.bed"]),
    (["density", "-i", "refs.bed", "reads_pos.bed"]),
    (["density", "-S", "-min", "0.05", "refs.bed", "reads_pos.bed"]),
]


@pytest.mark.parametrize("args", OVERLAP_RUNS, ids=[" ".join(a) for a in OVERLAP_RUNS])
def test_overlaps_cli_equals_oracle_cli(beds, args):
    want = oracle(args, cwd=beds)
    got = product("overlaps", args, cwd=beds)
    assert got[0] == want[0]
    assert got[1] == want[1]
    if want[0] != 0:
        assert got[2].strip() == want[2].strip()


def test_overlaps_stdin(beds):
    data = open(beds / "reads_pos.bed", "rb").read()
    want = oracle(["count", "-i", "refs.bed"], stdin=data, cwd=beds)
    got = product("overlaps", ["count", "-i", "refs.bed"], stdin=data, cwd=beds)
    assert got[:2] == want[:2]


SCAN_RUNS = [
    (["counts", "-i", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos.bed"]),
    (["counts", "-S", "-i", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos.bed"]),
    (["counts", "-g", "genome.bed", "-w", "500", "-d", "25", "-min", "3", "scan_strand.bed"]),
    (["counts", "-S", "-g", "genome.bed", "-w", "500", "-d", "25", "-min", "3", "scan_strand.bed"]),
    (["counts", "-i", "-op", "c", "-g", "genome.bed", "-w", "2000", "-d", "500", "-min", "2", "scan_pos.bed"]),
    (["counts", "-i", "-g", "genome.bed", "scan_pos.bed"]),                          # defaults -w 500 -d 25 -min 10
    (["counts", "-S", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos.bed"]),   # wrong order for strand-aware -S
    # -r: only windows that overlap a reference region (index of an in-memory set / merge with a sorted stream)
    (["counts", "-i", "-g", "genome.bed", "-r", "scan_refs_shuffled.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos.bed"]),
    (["counts", "-g", "genome.bed", "-r", "scan_refs_shuffled.bed", "-w", "500", "-d", "25", "-min", "2", "scan_strand.bed"]),
    (["counts", "-S", "-i", "-g", "genome.bed", "-r", "scan_refs.bed", "-w", "500", "-d", "100", "-min", "1", "scan_pos.bed"]),
    (["counts", "-i", "-Sref", "-g", "genome.bed", "-r", "scan_refs.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos.bed"]),
    (["counts", "-S", "-i", "-Sref", "-g", "genome.bed", "-r", "scan_refs.bed", "-w", "500", "-d", "25", "-min", "2", "scan_pos.bed"]),
    (["counts", "-Sref", "-g", "genome.bed", "-r", "scan_refs_strand.bed", "-w", "500", "-d", "25", "-min", "2", "scan_strand.bed"]),
    (["counts", "-i", "-Sref", "-g", "genome.bed", "-r", "scan_refs_shuffled.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos.bed"]),   # -Sref on unsorted regions
    (["counts", "-Sref", "-g", "genome.bed", "-r", "scan_refs.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_strand.bed"]),              # sorted by start, not by strand
]


@pytest.mark.parametrize("args", SCAN_RUNS, ids=[" ".join(a) for a in SCAN_RUNS])
def test_scans_cli_equals_oracle_cli(beds, args):
    want = oracle(args, cwd=beds)
    got = product("scans", args, cwd=beds)
    assert got[0] == want[0]
    if want[0] == 0:
        assert got[1] == want[1]
    else:
        assert got[2].strip() == want[2].strip()
        if "-Sref" in args:
            assert got[1] == want[1]                 # windows reported before the unsorted reference region was reached


# ---- G1: the reference's own example data, examples/genes.bed.gz (4785 overlapping,
# strand-mixed chr1 genes, space-separated BED6; copied as a data fixture) x 200k seeded 50 bp reads,
# in the five modes that must agree pairwise ------------------------------------------------------------
@pytest.fixture(scope="module")
def g1(tmp_path_factory):
    d = tmp_path_factory.mktemp("g1")
    rng = np.random.default_rng(101)
    n = 200_000
    s = np.sort(rng.integers(3_000_000, 197_000_000, size=n))
    strand = rng.choice(["+", "-"], size=n)
    tri = np.stack([np.zeros(n, dtype=np.int64), s + 1, s + 50], axis=1)
    write_bed(d / "reads_pos.bed", tri, ["chr1"], None, strand)
    o = np.lexsort((s, strand == "-"))
    write_bed(d / "reads_strand.bed", tri[o], ["chr1"], None, strand[o])
    # genes sorted the two ways -S needs (the shipped file is sorted by start)
    genes = gzip.open(os.path.join(GOLD, "genes.bed.gz"), "rt").read().splitlines()
    rows = [g.split() for g in genes]
    with open(d / "genes_pos.bed", "w") as f:
        f.write("\n".join(" ".join(r) for r in sorted(rows, key=lambda r: int(r[1]))) + "\n")
    with open(d / "genes_strand.bed", "w") as f:
        f.write("\n".join(" ".join(r) for r in sorted(rows, key=lambda r: (r[5] == "-", int(r[1])))) + "\n")
    return d


def test_g1_five_modes(g1):
    gz = os.path.join(GOLD, "genes.bed.gz")
    runs = {
        "unsorted": ["count", gz, "reads_pos.bed"],
        "unsorted_i": ["count", "-i", gz, "reads_pos.bed"],
        "S_i": ["count", "-S", "-i", "genes_pos.bed", "reads_pos.bed"],
        "S_s": ["count", "-S", "-s", "genes_strand.bed", "reads_strand.bed"],
        "S": ["count", "-S", "genes_pos.bed", "reads_pos.bed"],
    }
    out = {}
    for name, args in runs.items():
        want = oracle(args, cwd=g1)
        got = product("overlaps", args, cwd=g1)
        assert want[0] == 0 and got[0] == 0, (name, want[2], got[2])
        assert got[1] == want[1], name
        out[name] = dict(line.split("\t") for line in got[1].splitlines())
    # pairwise agreement inside each strand mode (two distinct results in all, as the survey observed)
    assert out["unsorted_i"] == out["S_i"]
    assert out["unsorted"] == out["S"] == out["S_s"]
    assert out["unsorted"] != out["unsorted_i"]


def test_g8_medium_single_chromosome(tmp_path):
    """1M reads x 200k refs on one chromosome as BED text through both CLIs."""
    refs = synth.refs_single_chrom(200_000, seed=42)
    reads = synth.reads_single_chrom(1_000_000, seed=42)
    write_bed(tmp_path / "refs.bed", refs, ["chr1"], ["e%d" % i for i in range(len(refs))])
    np.savetxt(tmp_path / "reads.bed", np.stack([reads[:, 1] - 1, reads[:, 2]], axis=1), fmt="chr1\t%d\t%d")
    for args in (["count", "-i", "refs.bed", "reads.bed"], ["count", "-S", "-i", "-min", "3", "refs.bed", "reads.bed"]):
        want = oracle(args, cwd=tmp_path)
        got = product("overlaps", args, cwd=tmp_path)
        assert got[:2] == want[:2]


# ---- sorted-merge corner cases: zero-length reads / regions, errors the packer must reproduce -----------
def test_sorted_mode_zero_length_regions(tmp_path):
    (tmp_path / "refs.bed").write_text("chr1\t100\t100\tZ\t0\t+\nchr1\t100\t200\tA\t0\t+\nchr1\t150\t150\tY\t0\t+\n")
    (tmp_path / "reads.bed").write_text("chr1\t50\t120\tr1\t0\t+\nchr1\t100\t100\tr2\t0\t+\nchr1\t100\t101\tr3\t0\t+\nchr1\t149\t151\tr5\t0\t+\nchr1\t150\t150\tr4\t0\t+\n")
    args = ["count", "-S", "-i", "refs.bed", "reads.bed"]
    want = oracle(args, cwd=tmp_path)
    got = product("overlaps", args, cwd=tmp_path)
    assert want[0] == 0
    assert got[:2] == want[:2]


@pytest.mark.parametrize("reads,frag", [
    ("chr1\t10\t20\tr\t0\t+\nchr1\t5\n", "number of tokens should be at least 3"),
    ("chr1\t10\t20\tr\t0\t+\nchr1\t30\t40\tr\t0\t*\n", "invalid strand '*'"),
    ("chr1\t10\t20\tr\t0\t+\nchr1\t-30\t-20\tr\t0\t+\n", "stop position must be positive"),
])
def test_ingest_errors_match_reference_text(tmp_path, reads, frag):
    (tmp_path / "refs.bed").write_text("chr1\t0\t1000\tA\t0\t+\n")
    (tmp_path / "reads.bed").write_text(reads)
    args = ["count", "-i", "refs.bed", "reads.bed"]
    want = oracle(args, cwd=tmp_path)
    got = product("overlaps", args, cwd=tmp_path)
    assert want[0] == 1 and got[0] == 1
    assert frag in want[2] and got[2].strip() == want[2].strip()


def test_no_gpu_free_paths_fail_loudly(tmp_path):
    (tmp_path / "refs.bed").write_text("chr1\t0\t1000\tA\t0\t+\n")
    rc, out, err = product("overlaps", ["subset", "refs.bed"], cwd=tmp_path)
    assert rc == 1 and "outside the MI355X" in err
    rc, out, err = product("overlaps", ["frobnicate", "refs.bed"], cwd=tmp_path)
    assert rc == 1 and "Unknown operation" in err
    rc, out, err = product("overlaps", ["count", "-Q", "refs.bed"], cwd=tmp_path)
    assert rc == 1 and "unknown option '-Q'" in err


def test_scans_reference_filter_keeps_exactly_the_overlapping_windows(beds):
    base = ["counts", "-i", "-g", "genome.bed", "-w", "1000", "-d", "500", "-min", "1"]
    allw = product("scans", base + ["scan_pos.bed"], cwd=beds)[1].splitlines()
    kept = product("scans", base + ["-r", "scan_refs_shuffled.bed", "scan_pos.bed"], cwd=beds)[1].splitlines()
    kept2 = product("scans", base + ["-Sref", "-r", "scan_refs.bed", "scan_pos.bed"], cwd=beds)[1].splitlines()
    assert kept == kept2 and 0 < len(kept) < len(allw)
    refs = {}
    for l in open(beds / "scan_refs.bed"):
        c, s, e = l.split("\t")[:3]
        refs.setdefault(c, []).append((int(s) + 1, int(e)))

    def overlaps(line):
        _, iv = line.split("\t")
        c, _, s, e = iv.split(" ")
        return any(rs <= int(e) and int(s) <= re for rs, re in refs.get(c, []))
    assert [l for l in allw if overlaps(l)] == kept


PEAK_RUNS = [
    (["peaks", "-i", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-g", "genome.bed", "-w", "1000", "-d", "250", "-min", "5", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-cmp", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-M", "poisson", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-cmp", "-M", "poisson", "-qval", "0.2", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-cmp", "-M", "binomial2", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-cmp", "-M", "cbinomial", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-cmp", "-M", "normal", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-norm", "-pval", "1e-4", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-g", "genome.bed", "-min", "6", "peaks_signal.bed", "peaks_control.bed"]),                     # strand-aware windows
    (["peaks", "-S", "-i", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),                      # sorted scanners, -op 1
    (["peaks", "-S", "-g", "genome.bed", "-min", "6", "peaks_signal_strand.bed", "peaks_control_strand.bed"]),
    (["peaks", "-i", "--max-label-value", "3", "-g", "genome.bed", "-min", "12", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-i", "-M", "bogus", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    # not sorted by strand: the scanners advance in lockstep, the error that is met first ends the run (here the control's, line 4,
    # 85 windows into chr1 '-'; the signal's line 5 would be met at window 190) -- behind the report lines with the whole files' counts
    (["peaks", "-S", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed"]),
    (["peaks", "-S", "-g", "genome.bed", "peaks_control.bed", "peaks_signal.bed"]),
]


@pytest.mark.parametrize("args", PEAK_RUNS, ids=[" ".join(a) for a in PEAK_RUNS])
def test_peaks_cli_equals_oracle_cli(beds, args):
    """genomic_scans peaks with a control: window counts from the GPU scans, tests on the host; p-values are printed
    with five digits, product and oracle use the same tail sums (tolerance parity with GSL)"""
    want = oracle(args, cwd=beds)
    got = product("scans", args, cwd=beds)
    assert got[0] == want[0]
    assert got[1] == want[1]
    assert got[2] == want[2]                           # the "* Effective genome size ..." report lines, the error if any
    if want[0] == 0:
        if "bogus" not in args:
            assert len(got[1].splitlines()) >= 10      # the planted clusters are found


@pytest.fixture(scope="module")
def tracks(beds):
    """mappability tracks in the scaled genome of `beds`, for the sorted scanner's operator 'p' (genomic_intervals.cpp:4939-4942)"""
    rng = np.random.default_rng(41)
    names = synth.CHROM_NAMES
    lens = synth.CHROM_LEN // 50
    def track(seed, n, gap_lo, gap_hi, len_lo, len_hi, strands=False, extra=None, disjoint=True):
        r = np.random.default_rng(seed)
        rows = []
        for c in np.argsort(np.array(names)):                                       # chromosomes in strcmp order
            for st in (("+", "-") if strands else ("+",)):
                pos = int(r.integers(1, 500))
                for _ in range(n):
                    ln = int(r.integers(len_lo, len_hi))
                    rows.append("%s\t%d\t%d\tm\t0\t%s" % (names[c], pos - 1, pos + ln - 1, st))
                    pos += int(r.integers(gap_lo, gap_hi)) + (ln if disjoint else 0)    # the gap to the next region, or the distance of the starts
                    if pos < 1: pos = 1
                    if pos > lens[c] + 2000: break
        if extra: extra(rows)
        return "\n".join(rows) + "\n"
    (beds / "uniq_disjoint.bed").write_text(track(1, 3000, 1, 4000, 5, 3000))            # regions that do not overlap
    (beds / "uniq_dense.bed").write_text(track(2, 4000, 0, 40, 1, 300))                  # short regions side by side: most of them end inside their micro-window, every other one is never looked at
    (beds / "uniq_overlapping.bed").write_text(track(3, 500, 1, 60, 20, 400, disjoint=False))   # sooner or later a moved start lies behind the next region's: the order error
    (beds / "uniq_strand.bed").write_text(track(4, 1500, 1, 5000, 5, 3000, strands=True))
    def bed12_mid(rows):                                                              # a spliced region in place of the middle row, same chromosome and start
        f = rows[len(rows) // 2].split("\t")
        rows[len(rows) // 2] = "%s\t%s\t%d\tx\t0\t+\t%s\t%d\t0\t2\t100,200\t0,600" % (f[0], f[1], int(f[1]) + 800, f[1], int(f[1]) + 800)
    (beds / "uniq_bed12.bed").write_text(track(5, 600, 1, 5000, 5, 3000, extra=bed12_mid))
    def unknown(rows):                                                                # two regions of a chromosome the genome file does not have, where they sort
        k = next(i for i, l in enumerate(rows) if l.startswith("chr2\t"))
        rows[k:k] = ["chr1_unplaced\t2\t300\tm\t0\t+", "chr1_unplaced\t5\t500\tm\t0\t+"]
    (beds / "uniq_unknown_chrom.bed").write_text(track(6, 800, 1, 5000, 5, 3000, extra=unknown))
    # lines the reader refuses, in the middle of a track: met when the walk PULLS them (as the region behind one it looks at, or as the
    # one its second pull steps over), with the windows in front of that point printed
    def bad_strand(rows): rows[len(rows) * 3 // 5] = rows[len(rows) * 3 // 5].rsplit("\t", 1)[0] + "\tx"
    (beds / "uniq_bad_strand.bed").write_text(track(7, 700, 1, 5000, 5, 3000, extra=bad_strand))
    def two_columns(rows): rows[len(rows) * 3 // 10] = "\t".join(rows[len(rows) * 3 // 10].split("\t")[:2])
    (beds / "uniq_two_columns.bed").write_text(track(8, 700, 1, 5000, 5, 3000, extra=two_columns))
    return beds


MAPPABILITY_RUNS = [
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "0", "uniq_disjoint.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "500", "-d", "25", "-min", "1", "uniq_disjoint.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "2000", "-d", "500", "-min", "0", "uniq_dense.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "300", "-d", "100", "-min", "0", "uniq_overlapping.bed"]),     # the order error of a moved start, windows in front of it printed
    (["counts", "-S", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "250", "-min", "1", "uniq_strand.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "250", "-min", "1", "uniq_strand.bed"]),           # '-' regions behind '+' ones: not sorted by position
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "500", "-min", "0", "uniq_bed12.bed"]),             # a spliced region: the error if the walk looks at it, nothing if it is skipped
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "200", "-min", "0", "uniq_bed12.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "500", "-min", "0", "uniq_unknown_chrom.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "500", "-min", "0", "uniq_bad_strand.bed"]),
    (["counts", "-S", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "250", "-min", "1", "uniq_two_columns.bed"]),
    (["counts", "-i", "-op", "p", "-g", "genome.bed", "-w", "1000", "-d", "500", "-min", "0", "uniq_disjoint.bed"]),                # the unsorted scanner refuses the operator
    (["peaks", "-i", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed", "uniq_disjoint.bed"]),
    (["peaks", "-S", "-i", "-g", "genome.bed", "-w", "1000", "-d", "250", "-min", "5", "peaks_signal.bed", "peaks_control.bed", "uniq_dense.bed"]),
    (["peaks", "-i", "-cmp", "-g", "genome.bed", "peaks_signal.bed", "peaks_control.bed", "uniq_disjoint.bed"]),
    (["peaks", "-g", "genome.bed", "-min", "6", "peaks_signal.bed", "peaks_control.bed", "uniq_strand.bed"]),
    (["peaks", "-i", "-g", "genome.bed", "-w", "300", "-d", "100", "peaks_signal.bed", "peaks_control.bed", "uniq_overlapping.bed"]),   # the track's order error ends the run, behind the report lines
]


@pytest.mark.parametrize("args", MAPPABILITY_RUNS, ids=[" ".join(a) for a in MAPPABILITY_RUNS])
def test_mappability_operator_equals_oracle_cli(tracks, args):
    """the sorted scanner's operator 'p' -- `genomic_scans counts -S -op p` and the third input of `peaks` -- with the walk the
    reference wrote (two pulls after a region that ends inside the micro-window, the moved start in the order check): stdout, stderr
    and exit code of the oracle CLI"""
    want = oracle(args, cwd=tracks)
    got = product("scans", args, cwd=tracks)
    assert got[0] == want[0], (got[2], want[2])
    assert got[1] == want[1]
    assert got[2].strip() == want[2].strip()


def test_mappability_track_of_a_million_regions(tmp_path):
    """the operator 'p' at the size of a real track: 1 M regions over 24 chromosomes of the hg38 lengths (strcmp order), two window
    geometries; stdout of the oracle CLI byte for byte (3.1 M and 12.4 M lines)"""
    rng = np.random.default_rng(77)
    names = np.array(synth.CHROM_NAMES)
    order = np.argsort(names)
    with open(tmp_path / "genome.bed", "w") as f:
        for c in order:
            f.write("%s\t0\t%d\n" % (names[c], synth.CHROM_LEN[c]))
    with open(tmp_path / "track.bed", "w") as f:
        for c in order:
            n = int(1_000_000 * synth.CHROM_LEN[c] / synth.CHROM_LEN.sum())
            gaps = rng.integers(1, 2 * int(synth.CHROM_LEN[c]) // n - 1200, size=n)
            lens = rng.integers(1, 2400, size=n)
            starts = np.cumsum(gaps + np.concatenate([[0], lens[:-1]]))
            keep = starts + lens < synth.CHROM_LEN[c]
            f.write("".join("%s\t%d\t%d\n" % (names[c], s, s + l) for s, l in zip(starts[keep], lens[keep])))
    for geom in (["-w", "1000", "-d", "1000"], ["-w", "1000", "-d", "250"]):
        args = ["counts", "-S", "-i", "-op", "p", "-g", "genome.bed"] + geom + ["-min", "0", "track.bed"]
        want = oracle(args, cwd=tmp_path)
        got = product("scans", args, cwd=tmp_path)
        assert got[0] == want[0] == 0, (got[2], want[2])
        assert got[1] == want[1] and len(got[1].splitlines()) > 3_000_000


def test_small_host_batches_give_the_same_output(beds, monkeypatch):
    """GTX_HOST_BATCH_READS=700: the input sets go to the device in dozens of batches through two buffers that are used in turn --
    totals that are summed over the batches (the label sums behind the background probability of `peaks`, the read count of `rpkm`)
    and everything else come out as with one batch"""
    runs = [("scans", PEAK_RUNS[0]), ("scans", PEAK_RUNS[2]), ("overlaps", ["count", "-i", "refs.bed", "reads_pos.bed"]),
            ("overlaps", ["count", "-S", "-i", "refs.bed", "reads_pos.bed"]), ("overlaps", ["rpkm", "-i", "refs.bed", "reads_shuffled.bed.gz"]),
            ("overlaps", ["coverage", "-i", "refs.bed", "reads_pos.bed"]), ("scans", SCAN_RUNS[0]), ("scans", SCAN_RUNS[3])]
    for tool, args in runs:
        cwd = beds
        one = product(tool, args, cwd=cwd)
        monkeypatch.setenv("GTX_HOST_BATCH_READS", "700")
        many = product(tool, args, cwd=cwd)
        monkeypatch.delenv("GTX_HOST_BATCH_READS")
        assert one[0] == many[0] == 0, (args, many[2])
        assert one[1] == many[1] and one[2] == many[2], args
        assert len(one[1].splitlines()) > 5, args


def test_partition_paths_through_the_tools(beds, monkeypatch):
    """GTX_BUCKET_MIN_READS=1: the tools' shuffled input takes the partition path of the library whatever its size (count: no sorted
    hint from the host side; coverage / density / scans: the library's own sample of the batch) -- byte for byte the oracle's output"""
    monkeypatch.setenv("GTX_BUCKET_MIN_READS", "1")
    for args in (["count", "-i", "refs.bed", "reads_shuffled.bed.gz"], ["count", "refs.bed", "reads_shuffled.bed.gz"],
                 ["coverage", "-i", "refs.bed", "reads_shuffled.bed.gz"], ["coverage", "-i", "--max-label-value", "4", "refs.bed", "reads_shuffled.bed.gz"],
                 ["density", "refs.bed", "reads_shuffled.bed.gz"], ["coverage", "-i", "-gaps", "refs.bed", "reads_shuffled.bed.gz"]):
        want = oracle(args, cwd=beds)
        got = product("overlaps", args, cwd=beds)
        assert got[0] == want[0] == 0, (args, got[2])
        assert got[1] == want[1], args
        assert len(got[1].splitlines()) > 5
    for args in (["counts", "-i", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "1", "reads_shuffled.bed.gz"],
                 ["counts", "-g", "genome.bed", "-w", "500", "-d", "25", "-min", "2", "reads_shuffled.bed.gz"]):
        want = oracle(args, cwd=beds)
        got = product("scans", args, cwd=beds)
        assert got[0] == want[0] == 0, (args, got[2])
        assert got[1] == want[1], args
        assert len(got[1].splitlines()) > 5


def test_peaks_finds_the_planted_clusters(beds):
    out = product("scans", ["peaks", "-i", "-g", "genome.bed", "-qval", "0.01", "peaks_signal.bed", "peaks_control.bed"], cwd=beds)[1].splitlines()
    assert len(out) > 100
    pv = [float(l.split("\t")[0]) for l in out]
    assert max(pv) < 1e-3


def test_peaks_without_control_runs(beds):
    import os as _os
    env = dict(_os.environ, GTX_SEED="5")
    r = subprocess.run([TOOLS["scans"], "peaks", "-i", "-g", "genome.bed", "peaks_signal.bed"], capture_output=True, cwd=beds, env=env)
    assert r.returncode == 0 and len(r.stdout.decode().splitlines()) > 50
    r2 = subprocess.run([TOOLS["scans"], "peaks", "-i", "-g", "genome.bed", "peaks_signal.bed"], capture_output=True, cwd=beds, env=env)
    assert r.stdout == r2.stdout                       # GTX_SEED fixes the background draws


# ---- packed region files (.gtx) as the streamed input: same output as the text they were made from -------------
PACKTOOL = os.path.join(BIN, "gtx_packtool")


@pytest.fixture(scope="module")
def packed(beds):
    for name in ("reads_pos", "reads_strand", "scan_pos", "scan_strand", "peaks_signal", "peaks_control"):
        r = subprocess.run([PACKTOOL, "pack", name + ".bed", name + ".gtx"], capture_output=True, cwd=beds)
        assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([PACKTOOL, "pack", "reads_shuffled.bed.gz", "reads_shuffled.gtx"], capture_output=True, cwd=beds)
    assert r.returncode == 0, r.stderr.decode()
    return beds


PACKED_RUNS = [
    ("overlaps", ["count", "-S", "-i", "refs.bed", "reads_pos"]),
    ("overlaps", ["count", "refs.bed", "reads_pos"]),
    ("overlaps", ["count", "-S", "-s", "refs_strand.bed", "reads_strand"]),
    ("overlaps", ["count", "-i", "--max-label-value", "4", "refs.bed", "reads_shuffled"]),
    ("overlaps", ["coverage", "-S", "-i", "refs.bed", "reads_pos"]),
    ("overlaps", ["rpkm", "-i", "refs.bed", "reads_pos"]),
    ("overlaps", ["count", "-S", "-i", "refs.bed", "reads_shuffled"]),        # not sorted: the same error either way
    ("scans", ["counts", "-i", "-g", "genome.bed", "-w", "1000", "-d", "1000", "-min", "1", "scan_pos"]),
    ("scans", ["counts", "-S", "-g", "genome.bed", "-w", "500", "-d", "25", "-min", "3", "scan_strand"]),
    ("scans", ["counts", "-i", "-g", "genome.bed", "-r", "scan_refs.bed", "-w", "1000", "-d", "500", "-min", "1", "scan_pos"]),
]


@pytest.mark.parametrize("tool,args", PACKED_RUNS, ids=[" ".join(a) for _, a in PACKED_RUNS])
def test_packed_input_equals_text_input(packed, tool, args):
    last = args[-1]
    text = product(tool, args[:-1] + [last + (".bed.gz" if last == "reads_shuffled" else ".bed")], cwd=packed)
    pk = product(tool, args[:-1] + [last + ".gtx"], cwd=packed)
    assert pk[0] == text[0] and pk[1] == text[1]
    if text[0] != 0:
        assert pk[2].strip() == text[2].strip()
    else:
        assert len(pk[1]) > 100


def test_packed_inputs_for_peaks(packed):
    base = ["peaks", "-i", "-g", "genome.bed", "-cmp"]
    text = product("scans", base + ["peaks_signal.bed", "peaks_control.bed"], cwd=packed)
    pk = product("scans", base + ["peaks_signal.gtx", "peaks_control.gtx"], cwd=packed)
    assert pk[0] == text[0] == 0 and pk[1] == text[1] and len(pk[1]) > 100
    assert pk[2].replace(".gtx", ".bed") == text[2]


# ---- sorted merge (-S): what the reference does with an index set that is out of order, and with inverted intervals ------------
def write_lines(path, rows):
    with open(path, "w") as f:
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")


@pytest.fixture(scope="module")
def merge_beds(tmp_path_factory):
    d = tmp_path_factory.mktemp("merge")
    rng = np.random.default_rng(71)
    names = ["chr1", "chr10", "chr2"]
    # index set sorted up to line 300 (0-based ordinal), then out of order; reads that stop before / go past that spot
    def regs(n, lo, hi, maxlen):
        c = np.sort(rng.integers(0, 3, size=n)); s = rng.integers(lo, hi, size=n); e = s + rng.integers(0, maxlen, size=n)
        o = np.lexsort((s, c))
        return c[o], s[o], e[o]
    c, s, e = regs(400, 1000, 900_000, 3000)
    st = rng.choice(["+", "-"], size=400)
    rows = [[names[c[i]], s[i], e[i], "g%d" % i, 0, st[i]] for i in range(400)]
    rows_bad = list(rows)
    rows_bad[300], rows_bad[301] = rows_bad[301], rows_bad[300]                      # disorder at ordinal 301 (class chr2)
    if rows_bad[300][1] == rows_bad[301][1]:
        rows_bad[301][1] -= 1
    write_lines(d / "refs_ok.bed", rows)
    write_lines(d / "refs_disorder.bed", rows_bad)
    qc, qs, qe = regs(20000, 1000, 900_000, 200)
    ql = rng.integers(0, 5, size=20000); qst = rng.choice(["+", "-"], size=20000)
    q = [[names[qc[i]], qs[i], qe[i], ql[i], 0, qst[i]] for i in range(20000)]
    write_lines(d / "reads_all.bed", q)                                               # reaches chr2: the merge gets to the bad spot
    stop_at = rows_bad[300][1]
    early = [r for r in q if r[0] != "chr2" or r[2] < min(rows[295][1], stop_at) - 5000]
    write_lines(d / "reads_early.bed", early)                                         # ends well before it: counts, exit 0
    write_lines(d / "reads_chr1.bed", [r for r in q if r[0] == "chr1"])
    write_lines(d / "reads_unknown_last.bed", [r for r in q if r[0] == "chr1"] + [["chrZ", 5, 9, 1, 0, "+"]])   # an unknown chromosome pulls everything
    # a query that is itself out of order BEFORE the merge reaches the bad spot: the query error wins
    qq = [r for r in q if r[0] != "chr2"]
    qq[100], qq[101] = qq[101], qq[100]
    if qq[100][1] == qq[101][1]:
        qq[101][1] -= 1
    write_lines(d / "reads_query_disorder.bed", qq + [r for r in q if r[0] == "chr2"])
    # inverted and zero-length intervals on both sides (BED start > end), sorted by start
    ic, is_, ie = regs(300, 1000, 200_000, 2000)
    irows = []
    for i in range(300):
        e2 = ie[i]
        if i % 17 == 0: e2 = is_[i] - int(rng.integers(1, 400))                     # inverted: BED end < start
        if i % 23 == 0: e2 = is_[i]                                                  # zero length: BED start == end
        irows.append([names[ic[i]], is_[i], e2, "h%d" % i, 0, "+-"[i % 2]])
    write_lines(d / "refs_inverted.bed", irows)
    jc, js, je = regs(30000, 500, 210_000, 300)
    jrows = []
    for i in range(30000):
        e2 = je[i]
        if i % 101 == 0: e2 = js[i] - int(rng.integers(1, 600))
        if i % 211 == 0: e2 = js[i]
        jrows.append([names[jc[i]], js[i], e2, int(rng.integers(0, 6)), 0, "+-"[int(rng.integers(0, 2))]])
    write_lines(d / "reads_inverted.bed", jrows)
    return d


MERGE_RUNS = [
    ["count", "-S", "-i", "refs_disorder.bed", "reads_early.bed"],          # disorder behind the last query: counts, rc 0
    ["count", "-S", "refs_disorder.bed", "reads_chr1.bed"],
    ["count", "-S", "-i", "refs_disorder.bed", "reads_all.bed"],            # the merge reaches it: the reference's error
    ["count", "-S", "-i", "refs_disorder.bed", "reads_unknown_last.bed"],   # a query on a chromosome behind all of the index pulls all of it
    ["count", "-S", "-i", "refs_disorder.bed", "reads_query_disorder.bed"], # the query error comes first
    ["coverage", "-S", "-i", "refs_disorder.bed", "reads_early.bed"],
    ["coverage", "-S", "-i", "refs_disorder.bed", "reads_all.bed"],
    ["count", "-S", "-i", "refs_ok.bed", "reads_all.bed"],
    ["count", "-S", "-i", "refs_inverted.bed", "reads_inverted.bed"],       # inverted intervals follow CalcDirection, no error
    ["count", "-S", "refs_inverted.bed", "reads_inverted.bed"],
    ["count", "-S", "-i", "--max-label-value", "4", "refs_inverted.bed", "reads_inverted.bed"],
    ["coverage", "-S", "-i", "refs_inverted.bed", "reads_inverted.bed"],
    ["coverage", "-S", "-i", "-gaps", "--max-label-value", "3", "refs_inverted.bed", "reads_inverted.bed"],
    ["coverage", "-S", "-gaps", "refs_inverted.bed", "reads_inverted.bed"],
    ["count", "-S", "-i", "-gaps", "refs_inverted.bed", "reads_inverted.bed"],
    ["density", "-S", "-i", "refs_inverted.bed", "reads_inverted.bed"],
    ["count", "-i", "refs_inverted.bed", "reads_early.bed"],                # bin index: invalid index regions are skipped
    ["count", "-i", "refs_ok.bed", "reads_inverted.bed"],                   # bin index: the inverted query is the reference's error
]


@pytest.mark.parametrize("args", MERGE_RUNS, ids=[" ".join(a) for a in MERGE_RUNS])
def test_sorted_merge_cli_equals_oracle_cli(merge_beds, args):
    want = oracle(args, cwd=merge_beds)
    got = product("overlaps", args, cwd=merge_beds)
    assert got[0] == want[0], (got[2], want[2])
    assert got[1] == want[1]
    assert got[2].strip() == want[2].strip()


def test_sorted_merge_cli_cases_are_what_they_claim(merge_beds):
    """The fixture really produces both outcomes of the lazy order check (so the parity above is not vacuous)."""
    rc, out, err = oracle(["count", "-S", "-i", "refs_disorder.bed", "reads_early.bed"], cwd=merge_beds)
    assert rc == 0 and len(out.splitlines()) == 400
    rc, out, err = oracle(["count", "-S", "-i", "refs_disorder.bed", "reads_all.bed"], cwd=merge_beds)
    assert rc == 1 and out == "" and "Line 301: index regions are not sorted" in err
    rc, out, err = oracle(["count", "-S", "-i", "refs_disorder.bed", "reads_query_disorder.bed"], cwd=merge_beds)
    assert rc == 1 and "query regions are not sorted" in err
    rc, out, err = oracle(["count", "-S", "-i", "refs_inverted.bed", "reads_inverted.bed"], cwd=merge_beds)
    assert rc == 0 and len(out.splitlines()) == 300


# ---- multi-interval (BED12) regions: matched on their envelopes under -gaps (genomic_intervals.cpp:5226, :5752, :5278) -----------
@pytest.fixture(scope="module")
def bed12(tmp_path_factory):
    from test_class_api import bed6, make_regions
    d = tmp_path_factory.mktemp("bed12")
    rng = np.random.default_rng(43)
    bed6(d / "refs12.bed", [r[:3] + ["b%d" % i] + r[4:] for i, r in enumerate(make_regions(rng, 2000, 2_000_000, 3000, bed12_frac=0.4))])
    bed6(d / "reads12.bed", make_regions(rng, 60000, 2_000_000, 400, bed12_frac=0.3))
    bed6(d / "reads12_shuffled.bed", make_regions(rng, 30000, 2_000_000, 400, bed12_frac=0.3, sort=False))
    bad = make_regions(rng, 50, 2_000_000, 400, bed12_frac=1.0)
    bad[20][10], bad[20][11] = "30,30,", "0,10,"                                   # blocks overlap: the region check fails
    bed6(d / "reads12_bad.bed", bad)
    bed6(d / "exons6.bed", [r[:3] + ["e%d" % i] + r[4:] for i, r in enumerate(make_regions(rng, 3000, 2_000_000, 300, bed12_frac=0.0))])
    bed6(d / "reads6.bed", make_regions(rng, 40000, 2_000_000, 200, bed12_frac=0.0))
    return d


BED12_RUNS = [
    ["count", "-i", "-gaps", "refs12.bed", "reads12.bed"],
    ["count", "-gaps", "refs12.bed", "reads12_shuffled.bed"],
    ["count", "-S", "-i", "-gaps", "refs12.bed", "reads12.bed"],
    ["count", "-S", "-gaps", "--max-label-value", "4", "refs12.bed", "reads12.bed"],
    ["coverage", "-i", "-gaps", "refs12.bed", "reads12.bed"],
    ["coverage", "-S", "-gaps", "--max-label-value", "3", "refs12.bed", "reads12.bed"],
    ["density", "-i", "-gaps", "refs12.bed", "reads12.bed"],
    ["rpkm", "-S", "-i", "-gaps", "refs12.bed", "reads12.bed"],
    ["count", "-i", "-gaps", "refs12.bed", "reads12_bad.bed"],                 # a region whose blocks overlap: the reference's error
    ["count", "-S", "-i", "-gaps", "refs12.bed", "reads12_bad.bed"],
]


@pytest.mark.parametrize("args", BED12_RUNS, ids=[" ".join(a) for a in BED12_RUNS])
def test_bed12_under_gaps_equals_oracle_cli(bed12, args):
    want = oracle(args, cwd=bed12)
    got = product("overlaps", args, cwd=bed12)
    assert got[0] == want[0], (got[2], want[2])
    assert got[1] == want[1]
    if want[0] != 0:
        assert got[2].strip() == want[2].strip()


# without -gaps: coverage / density are sums over ALL interval pairs of the two regions (CalcOverlap, genomic_intervals.cpp:1196-1202,
# :5278), so every interval goes to the device as a region / read of its own and a region's value is the sum of its intervals'
BED12_NOGAPS_RUNS = [
    ["coverage", "-i", "refs12.bed", "reads12.bed"],
    ["coverage", "refs12.bed", "reads12.bed"],
    ["coverage", "-i", "refs12.bed", "reads12_shuffled.bed"],
    ["coverage", "-S", "-i", "refs12.bed", "reads12.bed"],
    ["coverage", "-S", "--max-label-value", "3", "refs12.bed", "reads12.bed"],
    ["density", "-i", "refs12.bed", "reads12.bed"],
    ["density", "-S", "refs12.bed", "reads12.bed"],
    ["density", "-i", "exons6.bed", "reads12.bed"],                            # the shape of examples/example01.tcsh:15: spliced reads x exons
    ["density", "-i", "refs12.bed", "reads6.bed"],
    ["coverage", "-i", "refs12.bed", "reads12_bad.bed"],                       # a region whose blocks overlap: the reference's error
    ["coverage", "-S", "-i", "refs12.bed", "reads12_bad.bed"],
]


@pytest.mark.parametrize("args", BED12_NOGAPS_RUNS, ids=[" ".join(a) for a in BED12_NOGAPS_RUNS])
def test_bed12_coverage_without_gaps_equals_oracle_cli(bed12, args):
    want = oracle(args, cwd=bed12)
    got = product("overlaps", args, cwd=bed12)
    assert got[0] == want[0], (got[2], want[2])
    assert got[1] == want[1]
    if want[0] != 0:
        assert got[2].strip() == want[2].strip()
    elif args[-1] == "reads12.bed" and args[-2] == "refs12.bed":
        assert want[1] != oracle(args[:1] + ["-gaps"] + args[1:], cwd=bed12)[1]        # (the fixture tells the two rules apart)


# count without -gaps: a query counts once for an index region when their envelopes overlap and SOME interval of the one overlaps
# SOME interval of the other (genomic_intervals.cpp:1167-1172, :5226-5232) -- the streaming kernel counts on the envelopes, the pair
# kernels (csrc/gtx_pairs.hip) settle the pairs with a multi-interval side
@pytest.fixture(scope="module")
def genes12(tmp_path_factory):
    """Transcript-shaped regions (2-12 exons, introns up to 20 kb, a few regions that span most of a chromosome) and spliced reads."""
    from test_class_api import bed6, NAMES
    d = tmp_path_factory.mktemp("genes12")
    rng = np.random.default_rng(47)
    def regions(n, span, multi_frac, exon, intron, max_blocks, single_len, wide=0, sort=True):
        c = rng.integers(0, len(NAMES), size=n); s = rng.integers(0, span, size=n)
        if sort:
            o = np.lexsort((s, np.array([NAMES[i] for i in c]))); c, s = c[o], s[o]
        rows = []
        for i in range(n):
            strand = "+-"[int(rng.integers(0, 2))]; lab = int(rng.integers(0, 7))
            if i < wide or rng.random() >= multi_frac:
                ln = int(rng.integers(span // 2, span)) if i < wide else int(rng.integers(1, single_len))
                rows.append([NAMES[c[i]], int(s[i]), int(s[i]) + ln, lab, 0, strand])
                continue
            nb = int(rng.integers(2, max_blocks + 1)); sizes, starts, at = [], [], 0
            for _ in range(nb):
                starts.append(at); sz = int(rng.integers(exon[0], exon[1])); sizes.append(sz); at += sz + int(rng.integers(intron[0], intron[1]))
            end = int(s[i]) + starts[-1] + sizes[-1]
            rows.append([NAMES[c[i]], int(s[i]), end, lab, 0, strand, int(s[i]), end, 0, nb, ",".join(map(str, sizes)) + ",", ",".join(map(str, starts)) + ","])
        if sort:
            rows.sort(key=lambda r: (r[0], r[1]))
        return rows
    genes = regions(3000, 3_000_000, 0.7, (50, 300), (100, 20000), 12, 5000, wide=3)
    bed6(d / "genes12.bed", [r[:3] + ["t%d" % i] + r[4:] for i, r in enumerate(genes)])
    bed6(d / "spliced.bed", regions(80000, 3_000_000, 0.3, (10, 80), (50, 5000), 3, 120))
    bed6(d / "spliced_shuffled.bed", regions(40000, 3_000_000, 0.3, (10, 80), (50, 5000), 3, 120, sort=False))
    bed6(d / "unspliced.bed", regions(80000, 3_000_000, 0.0, None, None, 0, 120))
    bed6(d / "peaks6.bed", [r[:3] + ["p%d" % i] + r[4:] for i, r in enumerate(regions(4000, 3_000_000, 0.0, None, None, 0, 2000, wide=2))])
    # reads the sorted merge lets through and the bin index refuses or skips: zero-length (start == end), inverted (start > end),
    # and spliced reads with a block of size 0 -- all of them next to multi-interval regions
    odd = regions(20000, 3_000_000, 0.3, (10, 80), (50, 5000), 3, 120)
    for i, r in enumerate(odd):
        if len(r) == 6 and i % 7 == 0:
            r[2] = r[1]                                                              # zero length
        elif len(r) == 6 and i % 11 == 0:
            r[2] = max(r[1] - int(rng.integers(1, 400)), 0)                          # inverted
        elif len(r) == 12 and i % 5 == 0:
            sizes = r[10].rstrip(",").split(","); sizes[0] = "0"; r[10] = ",".join(sizes) + ","
    bed6(d / "odd_reads.bed", odd)
    bed6(d / "odd_reads_valid.bed", [r for r in odd if len(r) == 12 or r[2] > r[1]])  # (what the bin index accepts: start <= stop)
    by_strand = lambda rows: sorted(rows, key=lambda r: (r[0], r[5], r[1]))
    bed6(d / "genes12_strand.bed", by_strand([r[:3] + ["t%d" % i] + r[4:] for i, r in enumerate(genes)]))
    bed6(d / "spliced_strand.bed", by_strand(regions(50000, 3_000_000, 0.3, (10, 80), (50, 5000), 3, 120)))
    return d


BED12_COUNT_RUNS = [
    ("bed12", ["count", "-i", "refs12.bed", "reads12.bed"]),
    ("bed12", ["count", "refs12.bed", "reads12.bed"]),
    ("bed12", ["count", "-i", "refs12.bed", "reads12_shuffled.bed"]),
    ("bed12", ["count", "-S", "-i", "refs12.bed", "reads12.bed"]),
    ("bed12", ["count", "-S", "--max-label-value", "4", "refs12.bed", "reads12.bed"]),
    ("bed12", ["count", "-i", "exons6.bed", "reads12.bed"]),                          # spliced reads x single-interval regions
    ("bed12", ["count", "-i", "refs12.bed", "reads6.bed"]),                           # plain reads x multi-interval regions
    ("bed12", ["rpkm", "-S", "-i", "refs12.bed", "reads12.bed"]),
    ("bed12", ["count", "-i", "refs12.bed", "reads12_bad.bed"]),                      # a region whose blocks overlap: the reference's error
    ("bed12", ["count", "-S", "-i", "refs12.bed", "reads12_bad.bed"]),
    ("genes12", ["count", "-i", "genes12.bed", "spliced.bed"]),
    ("genes12", ["count", "genes12.bed", "spliced.bed"]),
    ("genes12", ["count", "-S", "-i", "genes12.bed", "spliced.bed"]),
    ("genes12", ["count", "-S", "-s", "genes12_strand.bed", "spliced_strand.bed"]),
    ("genes12", ["count", "-S", "-i", "--max-label-value", "5", "genes12.bed", "spliced.bed"]),
    ("genes12", ["count", "-i", "genes12.bed", "spliced_shuffled.bed"]),
    ("genes12", ["count", "-i", "genes12.bed", "unspliced.bed"]),
    ("genes12", ["count", "-S", "-i", "genes12.bed", "unspliced.bed"]),
    ("genes12", ["count", "-i", "peaks6.bed", "spliced.bed"]),
    ("genes12", ["count", "-S", "peaks6.bed", "spliced.bed"]),
    ("genes12", ["rpkm", "-i", "genes12.bed", "spliced.bed"]),
    ("genes12", ["count", "-S", "-i", "genes12.bed", "odd_reads.bed"]),               # zero-length / inverted reads, blocks of size 0
    ("genes12", ["count", "-S", "genes12.bed", "odd_reads.bed"]),
    ("genes12", ["count", "-S", "-i", "peaks6.bed", "odd_reads.bed"]),
    ("genes12", ["count", "-i", "genes12.bed", "odd_reads_valid.bed"]),
    ("genes12", ["count", "-i", "genes12.bed", "odd_reads.bed"]),                     # the bin index's error for start > stop
]


@pytest.mark.parametrize("where,args", BED12_COUNT_RUNS, ids=[" ".join(a) for _, a in BED12_COUNT_RUNS])
def test_bed12_count_without_gaps_equals_oracle_cli(request, where, args):
    cwd = request.getfixturevalue(where)
    want = oracle(args, cwd=cwd)
    got = product("overlaps", args, cwd=cwd)
    assert got[0] == want[0], (got[2], want[2])
    assert got[1] == want[1]
    if want[0] != 0:
        assert got[2].strip() == want[2].strip()
    elif "bad" not in args[-1] and not (args[-1].startswith("odd") and args[-2] == "peaks6.bed"):
        assert want[1] != oracle(args[:1] + ["-gaps"] + args[1:], cwd=cwd)[1]          # (the fixture tells the two rules apart)


@pytest.mark.parametrize("args", [["count", "-i", "genes12.bed", "unspliced.bed"], ["count", "-S", "-i", "genes12.bed", "spliced.bed"],
                                  ["count", "-S", "genes12.bed", "spliced.bed"]], ids=lambda a: " ".join(a))
def test_bed12_count_with_text_on_device(genes12, args):
    """The query text tokenised on the device: plain lines are checked against the multi-interval regions where their triples are, a
    block with 12-column lines goes back to the host packer, which lists their intervals."""
    want = oracle(args, cwd=genes12)
    rc, out, err, nums = _product_text(args, genes12, block_mb=1)
    assert rc == want[0] == 0, err
    assert out == want[1]
    assert nums is not None and sum(nums) > 0
    if args[-1] == "unspliced.bed":
        assert nums[0] > 0 and nums[1] == 0                                             # every block stayed on the device


def test_bed12_count_in_memory_query_set(genes12):
    """A caller that holds the query set in memory (load_in_memory = true): its multi-interval regions reach the packer as 12-column lines."""
    caller = os.path.join(BIN, "api_caller")
    for opts in (["-i"], ["-S", "-i"]):
        want = oracle(["count"] + opts + ["genes12.bed", "spliced.bed"], cwd=genes12)
        r = subprocess.run([caller, "icount"] + opts + ["genes12.bed", "spliced.bed"], capture_output=True, cwd=genes12)
        assert r.returncode == 0 == want[0], r.stderr.decode()
        assert r.stdout.decode() == want[1]


# ---- BED text tokenised on the device (gtx_count_add_text, csrc/gtx_text.hip): the plain case there, everything else back to the host packer
def _product_text(args, cwd, block_mb=None):
    env = dict(os.environ, GTX_TEXT_ON_DEVICE="1", GTX_TEXT_TRACE="1")
    if block_mb:
        env["GTX_PACK_BLOCK_MB"] = str(block_mb)
    r = subprocess.run([TOOLS["overlaps"]] + list(args), capture_output=True, cwd=cwd, env=env)
    lines = r.stderr.decode().split("\n")                                          # (not splitlines: a '\r' may be part of a message)
    trace = [l for l in lines if l.startswith("[gtx text]")]
    err = "\n".join(l for l in lines if not l.startswith("[gtx text]"))
    nums = [int(x) for x in __import__("re").findall(r": (\d+)", trace[0])] if trace else None
    return r.returncode, r.stdout.decode(), err, nums


@pytest.fixture(scope="module")
def text_beds(tmp_path_factory):
    d = tmp_path_factory.mktemp("textdev")
    rng = np.random.default_rng(77)
    names = ["chr1", "chr10", "chr2", "chrX"]
    def lines(n, lo, hi, width, sort=True, extra_chrom=None):
        c = rng.integers(0, len(names), n); s = rng.integers(lo, hi, n); ln = rng.integers(20, width, n)
        if sort:
            o = np.lexsort((s, c)); c, s, ln = c[o], s[o], ln[o]
        out = []
        for i in range(n):
            nm = names[c[i]] if extra_chrom is None or i % 97 else extra_chrom
            out.append("%s\t%d\t%d\tr%d\t%d\t%s" % (nm, s[i], s[i] + ln[i], rng.integers(0, 9), 0, "+-"[int(rng.integers(0, 2))]))
        return out
    (d / "refs.bed").write_text("\n".join(lines(3000, 1000, 2_000_000, 3000)) + "\n")
    plain = lines(200_000, 1000, 2_000_000, 300)
    (d / "plain.bed").write_text("\n".join(plain) + "\n")
    (d / "unknown_chrom.bed").write_text("\n".join(lines(100_000, 1000, 2_000_000, 300, extra_chrom="chr1_random")) + "\n")   # dropped lines; also breaks the order for -S
    (d / "shuffled.bed").write_text("\n".join(lines(120_000, 1000, 2_000_000, 300, sort=False)) + "\n")
    def variant(name, edit, at=150_000):
        v = list(plain); v[at] = edit(v[at]); (d / name).write_text("\n".join(v) + "\n")
    variant("crlf.bed", lambda l: l + "\r")
    variant("spaces.bed", lambda l: l.replace("\t", " "))
    variant("plus_sign.bed", lambda l: l.split("\t")[0] + "\t+" + "\t".join(l.split("\t")[1:]))
    variant("bad_strand.bed", lambda l: "\t".join(l.split("\t")[:5] + ["x"]))
    variant("three_cols.bed", lambda l: "\t".join(l.split("\t")[:3]))
    variant("two_cols.bed", lambda l: "\t".join(l.split("\t")[:2]))
    variant("empty_token.bed", lambda l: l.replace("\t0\t", "\t\t"))
    variant("bed12.bed", lambda l: l + "\t0\t0\t0\t1\t%d,\t0," % (int(l.split("\t")[2]) - int(l.split("\t")[1])))
    variant("zero_stop.bed", lambda l: "chr1\t0\t0\tz\t0\t+", at=0)
    variant("inverted.bed", lambda l: "\t".join([l.split("\t")[0], l.split("\t")[2], l.split("\t")[1]] + l.split("\t")[3:]))
    variant("label_text.bed", lambda l: "\t".join(l.split("\t")[:3] + ["7up"] + l.split("\t")[4:]))
    v = list(plain)
    v[150_000] = "chr1\t1999999\t2000100\tlate\t0\t+"                                      # out of order inside a block
    (d / "disorder.bed").write_text("\n".join(v) + "\n")
    (d / "no_final_newline.bed").write_text("\n".join(plain))
    return d


TEXT_RUNS = [["count", "-S", "-i"], ["count", "-i"], ["count", "-S"], ["count"], ["count", "-S", "-s"], ["count", "-S", "-i", "--max-label-value", "5"],
             ["coverage", "-S", "-i"], ["coverage", "-i", "--max-label-value", "3"], ["density", "-S"], ["rpkm", "-S", "-i"]]


@pytest.mark.parametrize("mode", TEXT_RUNS, ids=[" ".join(m) for m in TEXT_RUNS])
def test_text_on_device_plain_file(text_beds, mode):
    """a plain tab-separated BED file: every block is tokenised on the device, none comes back, output = the oracle CLI's"""
    reads = "plain.bed"
    if mode[-1] == "-s":                                                             # sorted by strand: regroup the file
        rows = sorted((l.split("\t") for l in (text_beds / "plain.bed").read_text().splitlines()), key=lambda r: (r[0], r[5], int(r[1])))
        (text_beds / "plain_by_strand.bed").write_text("\n".join("\t".join(r) for r in rows) + "\n")
        reads = "plain_by_strand.bed"
        rows = sorted((l.split("\t") for l in (text_beds / "refs.bed").read_text().splitlines()), key=lambda r: (r[0], r[5], int(r[1])))
        (text_beds / "refs_by_strand.bed").write_text("\n".join("\t".join(r) for r in rows) + "\n")
    args = mode + ["refs_by_strand.bed" if mode[-1] == "-s" else "refs.bed", reads]
    want = oracle(args, cwd=text_beds)
    rc, out, err, nums = _product_text(args, text_beds, block_mb=1)
    assert (rc, out) == (want[0], want[1]), err
    assert nums is not None and nums[0] >= 5 and nums[1] == 0 and nums[2] == 0, nums


ODD_FILES = ["crlf.bed", "spaces.bed", "plus_sign.bed", "bad_strand.bed", "three_cols.bed", "two_cols.bed", "empty_token.bed", "bed12.bed", "zero_stop.bed",
             "inverted.bed", "label_text.bed", "disorder.bed", "no_final_newline.bed", "unknown_chrom.bed", "shuffled.bed"]


@pytest.mark.parametrize("name", ODD_FILES)
def test_text_on_device_odd_lines_go_back_to_the_host(text_beds, name):
    """one line outside the plain case (or an error of the reference's) in a file of 200 k lines: that block is redone by the host
    packer -- output, exit code and message are the oracle CLI's in both algorithms; the other blocks stay on the device"""
    gaps = ["-gaps"] if name == "bed12.bed" else []          # (a COUNT over multi-interval regions is inside the path under -gaps only)
    for mode in (["count", "-S", "-i"] + gaps, ["count", "-i", "--max-label-value", "9"] + gaps, ["coverage", "-i"]):
        args = mode + ["refs.bed", name]
        want = oracle(args, cwd=text_beds)
        rc, out, err, nums = _product_text(args, text_beds, block_mb=1)
        assert rc == want[0], (args, err, want[2])
        assert out == want[1], args
        if want[0] != 0:
            assert err.strip() == want[2].strip(), args
        elif name in ("crlf.bed", "spaces.bed", "plus_sign.bed", "empty_token.bed", "bed12.bed"):
            assert nums is not None and nums[1] >= 1, (args, nums)                    # the odd line's block did come back (odd under either algorithm's rules)
        if name in ("three_cols.bed", "label_text.bed", "no_final_newline.bed") and want[0] == 0:
            assert nums is not None and nums[1] == 0, (args, nums)                    # plain after all: a 3-column line, a label atol reads, a dropped tail


def _run_text(tool, args, cwd, stdin=None, block_mb=1, extra_env=None):
    env = dict(os.environ, GTX_TEXT_ON_DEVICE="1", GTX_TEXT_TRACE="1", GTX_PACK_BLOCK_MB=str(block_mb))
    env.update(extra_env or {})
    r = subprocess.run([TOOLS[tool]] + list(args), capture_output=True, cwd=cwd, env=env, input=stdin)
    lines = r.stderr.decode().split("\n")
    trace = [l for l in lines if l.startswith("[gtx text]")]
    nums = [int(x) for x in __import__("re").findall(r": (\d+)", trace[0])] if trace else None
    return r.returncode, r.stdout.decode(), "\n".join(l for l in lines if not l.startswith("[gtx text]")), nums


@pytest.mark.parametrize("mode", [["count", "-S", "-i"], ["count", "-i", "--max-label-value", "5"], ["coverage", "-S", "-i"], ["density", "-i"]],
                         ids=lambda m: " ".join(m))
def test_text_on_device_from_a_pipe_and_from_gz(text_beds, mode):
    """The inputs the reference's examples use (examples/example01.tcsh:15: `cat reads | genomic_overlaps density -v exons.bed`; every
    shipped data file is .gz): stdin and a gzip file are read block by block like a regular file -- straight reads, or inflate, into
    the page-locked buffers -- and tokenised on the device; output = the oracle CLI's on the plain file."""
    import gzip
    want = oracle(mode + ["refs.bed", "plain.bed"], cwd=text_beds)
    text = (text_beds / "plain.bed").read_bytes()
    rc, out, err, nums = _run_text("overlaps", mode + ["refs.bed"], text_beds, stdin=text)
    assert (rc, out) == (want[0], want[1]), err
    assert nums is not None and nums[0] >= 5 and nums[1] == 0 and nums[2] == 0, nums
    gz = text_beds / "plain.bed.gz"
    if not gz.exists():
        gz.write_bytes(gzip.compress(text, 1))
    rc, out, err, nums = _run_text("overlaps", mode + ["refs.bed", "plain.bed.gz"], text_beds)
    assert (rc, out) == (want[0], want[1]), err
    assert nums is not None and nums[0] >= 5 and nums[1] == 0 and nums[2] == 0, nums
    # a pipe whose last line has no newline (dropped, core.cpp:243) and one with an odd line in the middle (that block back to the host)
    rc, out, err, nums = _run_text("overlaps", mode + ["refs.bed"], text_beds, stdin=text[:-1])
    want2 = oracle(mode + ["refs.bed", "no_final_newline.bed"], cwd=text_beds)
    assert (rc, out) == (want2[0], want2[1]), err
    rc, out, err, nums = _run_text("overlaps", mode + ["refs.bed"], text_beds, stdin=(text_beds / "spaces.bed").read_bytes())
    want3 = oracle(mode + ["refs.bed", "spaces.bed"], cwd=text_beds)
    assert (rc, out) == (want3[0], want3[1]) and nums[1] >= 1, (err, nums)


SCAN_TEXT_RUNS = [["counts", "-i", "-w", "1000", "-d", "1000", "-min", "1"], ["counts", "-w", "500", "-d", "25", "-min", "2"],
                  ["counts", "-S", "-i", "-w", "2000", "-d", "1000", "-min", "1"], ["counts", "-i", "-op", "c", "-w", "1000", "-d", "500", "-min", "1"],
                  ["counts", "-i", "--max-label-value", "4", "-w", "1000", "-d", "1000", "-min", "1"]]


@pytest.mark.parametrize("mode", SCAN_TEXT_RUNS, ids=lambda m: " ".join(m))
def test_text_on_device_genomic_scans(text_beds, mode):
    """genomic_scans counts fed as a stream (gtx_scan_begin .. gtx_scan_end), its input tokenised on the device: a plain file, a pipe,
    and files with lines the scanners skip (start > stop, stop = 0, unknown chromosome) or the device hands back (spaces, \\r);
    the sorted scanner's order error comes from the host packer with the reference's text."""
    (text_beds / "genome.bed").write_text("chr1\t0\t2100000\nchr10\t0\t2100000\nchr2\t0\t1500000\nchrX\t0\t2100000\n")
    g = ["-g", "genome.bed"]
    for name in ("plain.bed", "inverted.bed", "zero_stop.bed", "unknown_chrom.bed", "spaces.bed", "crlf.bed", "disorder.bed"):
        if "-S" in mode and name in ("unknown_chrom.bed",):
            continue                                                         # (chr1_random lines break the order: covered by disorder.bed)
        want = oracle(mode + g + [name], cwd=text_beds)
        rc, out, err, nums = _run_text("scans", mode + g + [name], text_beds)
        assert rc == want[0], (name, err, want[2])
        # (the sorted scanner streams: by the time it meets an offending line the reference has printed the windows in front of the
        # region before that line -- the same bytes here, then the same message and exit code)
        assert out == want[1], name
        if want[0] != 0:
            assert err.strip() == want[2].strip(), name
        elif name in ("plain.bed", "inverted.bed", "zero_stop.bed") and "-S" not in mode:
            assert nums is not None and nums[0] >= 5 and nums[1] == 0, (name, nums)
    want = oracle(mode + g + ["plain.bed"], cwd=text_beds)
    rc, out, err, nums = _run_text("scans", mode + g, text_beds, stdin=(text_beds / "plain.bed").read_bytes())
    assert (rc, out) == (want[0], want[1]) and nums[0] >= 5, (err, nums)


def test_text_on_device_several_gpus(text_beds):
    """--ngpu 3 (rehearsed on one device): the blocks of the query file go to the members in turn (gtx_group_count_add_text), whatever
    their chromosomes, a block with an odd line comes back and is routed by class like any packed batch, and the call ends with the
    sum of the members' full vectors"""
    for mode in (["count", "-S", "-i"], ["count", "-i", "--max-label-value", "5"], ["coverage", "-i"]):
        for name in ("plain.bed", "spaces.bed", "shuffled.bed"):
            if "-S" in mode and name == "shuffled.bed":
                continue
            want = oracle(mode + ["refs.bed", name], cwd=text_beds)
            rc, out, err, nums = _run_text("overlaps", [mode[0], "--ngpu", "3"] + mode[1:] + ["refs.bed", name], text_beds, extra_env={"GTX_GROUP_REHEARSE": "1"})
            assert (rc, out) == (want[0], want[1]), (mode, name, err)
            assert nums is not None and nums[0] >= 3, (mode, name, nums)


SORTED_SCAN_ERRORS = [
    ("late_disorder", lambda v: v.__setitem__(150_000, "chr1\t1999999\t2000100\tlate\t0\t+")),             # out of order in the middle of chr10
    ("first_line_bad", lambda v: v.__setitem__(0, "chr1\t100\t200\tx\t0\tq")),                              # the constructor's first read meets it: nothing printed
    ("second_line_bad", lambda v: v.__setitem__(1, "chr1\t100")),                                            # met when the first region is consumed
    ("bad_strand_late", lambda v: v.__setitem__(199_990, "\t".join(v[199_990].split("\t")[:5] + ["x"]))),     # on chrX, near the end
    ("behind_the_bounds", lambda v: v.extend(["chrX\t2100500\t2100600\tedge\t0\t+", "chrX\t5\t10\tback\t0\t+"])),   # the region before the bad line starts behind chrX's last micro-window: the head of no later block consumes it
    ("unknown_chrom_then_bad", lambda v: (v.__setitem__(100_000, "chr1_zzz\t5\t10\tu\t0\t+"), v.__setitem__(100_001, "chr2\t1\t2\tb\t0\tq"))),   # the region before the bad line has no bounds: the skip loop at the head of chr2's block consumes it
]


@pytest.mark.parametrize("name,edit", SORTED_SCAN_ERRORS, ids=[n for n, _ in SORTED_SCAN_ERRORS])
def test_sorted_scanner_meets_errors_where_the_reference_does(text_beds, name, edit):
    """genomic_scans counts -S streams in the reference: an error of the input is met when the region in FRONT of the offending line is
    consumed, with the windows before that point already printed -- or never, when that region lies behind every block.  Output up to
    the error, the message and the exit code must be the oracle CLI's, from the host packer and from the device tokenizer alike, with
    and without strands, at two geometries."""
    (text_beds / "genome.bed").write_text("chr1\t0\t2100000\nchr10\t0\t2100000\nchr2\t0\t1500000\nchrX\t0\t2100000\n")
    v = (text_beds / "plain.bed").read_text().splitlines()
    edit(v)
    (text_beds / ("sse_%s.bed" % name)).write_text("\n".join(v) + "\n")
    for mode in (["counts", "-S", "-i", "-w", "1000", "-d", "1000", "-min", "0"], ["counts", "-S", "-i", "-w", "2000", "-d", "500", "-min", "1"]):
        args = mode + ["-g", "genome.bed", "sse_%s.bed" % name]
        want = oracle(args, cwd=text_beds)
        for dev in ("0", "1"):
            rc, out, err, nums = _run_text("scans", args, text_beds, extra_env={"GTX_TEXT_ON_DEVICE": dev})
            assert rc == want[0], (name, mode, dev, err, want[2])
            assert out == want[1], (name, mode, dev, len(out), len(want[1]))
            assert err.strip() == want[2].strip(), (name, mode, dev)
