"""CPU suite, part 4: the C++ BED ingest (gtx_bed.*) seen through gtx_packtool -- no GPU involved.
Expected triples come from a straightforward Python reading of the same rules (tab-or-space separator,
start = col2 + 1, class = strcmp rank [+ n_chrom for '-'], '-' reads grouped behind '+' reads)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd", "csrc", "gtx_packtool")
GOLD = os.path.join(ROOT, "tests", "golden")


def pack(args, stdin=None, cwd=None):
    r = subprocess.run([TOOL] + list(args), input=stdin, capture_output=True, cwd=cwd)
    rows = [tuple(int(x) for x in l.split()) for l in r.stdout.decode().splitlines() if l and not l.startswith("#")]
    meta = [l for l in r.stdout.decode().splitlines() if l.startswith("#")]
    return r.returncode, rows, meta, r.stderr.decode()


def test_toy_strand_aware_grouping():
    rc, rows, meta, _ = pack(["ou", "-a", "-c", "chr1,chr2", "g2_reads.bed"], cwd=GOLD)
    assert rc == 0
    # chr3 is unknown -> dropped; the '-' read (class 2 = 0 + n_chrom) comes after all '+' reads
    assert rows == [(0, 51, 100), (0, 51, 101), (0, 200, 250), (0, 201, 250), (1, 20, 30), (2, 300, 350)]
    assert meta[-1] == "# lines=7"


@pytest.mark.parametrize("name,expect", [("g5_reads_no_final_newline.bed", 2), ("g5_reads_space_separated.bed", 2),
                                         ("g5_reads_with_header.bed", 2), ("g5_reads.bed.gz", 6)])
def test_ingest_quirks(name, expect):
    # headers are skipped by GenomicRegionSet (not by the packer): feed the tool only files without them
    if "header" in name:
        data = b"".join(l for l in open(os.path.join(GOLD, name), "rb").readlines() if not l.startswith((b"track", b"browser")))
        rc, rows, meta, err = pack(["ou", "-c", "chr1,chr2"], stdin=data)
    else:
        rc, rows, meta, err = pack(["ou", "-c", "chr1,chr2", name], cwd=GOLD)
    assert rc == 0, err
    assert len(rows) == expect


def test_labels_become_clamped_weights():
    rc, rows, _, _ = pack(["ou", "-l", "3", "-c", "chr1", "g3_reads.bed"], cwd=GOLD)
    assert rc == 0 and [r[3] for r in rows] == [3, 2, 0]


@pytest.mark.parametrize("mode,file,frag", [
    ("ou", "g4_reads_zero_length.bed", "Error: Line 2: start position cannot be greater than stop position!"),
    ("os", "g4_reads_zero_length.bed", None),                       # the sorted merge has no such check
])
def test_degenerate_read_rules(mode, file, frag):
    rc, rows, meta, err = pack([mode, "-z", "-c", "chr1", file], cwd=GOLD)
    if frag:
        assert rc == 1 and frag in err
    else:
        assert rc == 0 and (0, 151, 150) in rows and "# zero 0 151 1" in meta


def make_bed(n, seed, sort=True):
    rng = np.random.default_rng(seed)
    names = ["chr1", "chr10", "chr2", "chrX"]
    c = rng.integers(0, 4, size=n)
    s = rng.integers(0, 5_000_000, size=n)
    ln = rng.integers(1, 500, size=n)
    st = rng.choice(["+", "-"], size=n)
    lab = rng.integers(0, 9, size=n)
    if sort:
        o = np.lexsort((s, c))
        c, s, ln, st, lab = c[o], s[o], ln[o], st[o], lab[o]
    text = "".join("%s\t%d\t%d\t%d\t0\t%s\n" % (names[ci], si, si + li, la, sti) for ci, si, li, la, sti in zip(c, s, ln, lab, st))
    rows = [(int(ci), int(si) + 1, int(si + li), int(la), sti) for ci, si, li, la, sti in zip(c, s, ln, lab, st)]
    return text.encode(), rows


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_many_lines_any_thread_count(tmp_path, threads):
    data, rows = make_bed(120_000, 5)
    (tmp_path / "r.bed").write_bytes(data)
    rc, got, meta, err = pack(["os", "-a", "-l", "5", "-t", str(threads), "-b", "50000", "-c", "chr1,chr10,chr2,chrX", "r.bed"], cwd=tmp_path)
    assert rc == 0, err
    want = [(c + (4 if st == "-" else 0), s, e, min(5, la)) for c, s, e, la, st in rows]
    # inside every batch the '+' reads come first, then the '-' reads, each in file order: compare as multisets
    # and check that every class's reads keep their file order
    assert sorted(got) == sorted(want)
    for cls in range(8):
        assert [g for g in got if g[0] == cls] == [w for w in want if w[0] == cls]
    assert meta[-1] == "# lines=120000"


def test_gz_and_stdin_agree_with_file(tmp_path):
    data, _ = make_bed(30_000, 6)
    (tmp_path / "r.bed").write_bytes(data)
    with gzip.open(tmp_path / "r.bed.gz", "wb") as f:
        f.write(data)
    a = pack(["ou", "-c", "chr1,chr2", "r.bed"], cwd=tmp_path)
    b = pack(["ou", "-c", "chr1,chr2", "r.bed.gz"], cwd=tmp_path)
    c = pack(["ou", "-c", "chr1,chr2"], stdin=data)
    assert a[0] == 0 and a[1] == b[1] == c[1] and len(a[1]) > 0


def test_order_violation_reports_the_reference_line(tmp_path):
    data, rows = make_bed(50_000, 7)
    lines = data.split(b"\n")
    lines[31_000], lines[31_001] = lines[31_001], lines[31_000]          # swap two neighbours with different starts
    bad = b"\n".join(lines)
    for threads in (1, 7):
        rc, _, _, err = pack(["os", "-t", str(threads), "-c", "chr1,chr10,chr2,chrX"], stdin=bad)
        s0, s1 = rows[31_000], rows[31_001]
        if (s0[0], s0[1]) == (s1[0], s1[1]):
            assert rc == 0
        else:
            assert rc == 1 and "Error: Line 31002: query regions are not sorted (sorted-by-strand = false)!" in err
    # the unsorted rules accept the same input
    assert pack(["ou", "-c", "chr1,chr10,chr2,chrX"], stdin=bad)[0] == 0


def test_first_error_in_file_order_wins(tmp_path):
    data, _ = make_bed(40_000, 8)
    lines = data.split(b"\n")
    lines[35_000] = b"chr1\t5"                                             # too few tokens, late
    lines[12_345] = b"chr1\t10\t20\tx\t0\t?"                               # bad strand, early
    rc, _, _, err = pack(["ou", "-t", "8", "-c", "chr1,chr10,chr2,chrX"], stdin=b"\n".join(lines))
    assert rc == 1 and "invalid strand '?'" in err and "tokens" not in err


def test_bed12_and_range_are_rejected():
    rc, _, _, err = pack(["ou", "-c", "chr1"], stdin=b"chr1\t0\t100\tx\t0\t+\t0\t100\t0\t2\t10,10\t0,50\n")
    assert rc == 1 and "BED12" in err
    rc, _, _, err = pack(["ou", "-c", "chr1"], stdin=b"chr1\t0\t3000000000\n")
    assert rc == 1 and "32-bit" in err


@pytest.mark.parametrize("mode,extra", [("ou", []), ("os", ["-z"]), ("su", ["-a"]), ("ss", ["-a", "-l", "3"]), ("ou", ["-a", "-l", "4"])])
def test_one_pass_tab_parser_agrees_with_the_general_tokenizer(tmp_path, mode, extra):
    """The fast path for plain TAB-separated lines must be indistinguishable from the tokenizer that follows the
    reference's rules: random lines built from digits, signs, letters, dots and TABs (no blanks), 1..13 columns,
    empty columns, trailing TABs, odd strand tokens -- same triples, weights, line counts and the same first error."""
    rng = np.random.default_rng(len(mode) * 100 + len(extra))
    alphabet = list("0123456789") * 3 + list("-+.ax1")
    for trial in range(int(os.environ.get("GTX_PARSE_TRIALS", "60"))):
        lines = []
        for _ in range(int(rng.integers(1, 40))):
            ncol = int(rng.choice([1, 2, 3, 3, 4, 5, 6, 6, 6, 7, 11, 12, 13]))
            cols = []
            for c in range(ncol):
                if c == 0:
                    cols.append(str(rng.choice(["chr1", "chr2", "chr9", "chrX", ""])))
                elif c in (1, 2) and rng.random() < 0.85:
                    kind = int(rng.integers(0, 8))                         # the digit fast path: 1..10 plain digits; everything else the loop
                    if kind < 3: cols.append(str(int(rng.integers(-5, 5000))))
                    elif kind == 3: cols.append(str(int(rng.integers(10**6, 10**8))))
                    elif kind == 4: cols.append(str(int(rng.integers(10**8, 2147483640))))
                    elif kind == 5: cols.append("0" * int(rng.integers(1, 7)) + str(int(rng.integers(0, 99999))))
                    elif kind == 6: cols.append(str(rng.choice(["+", "-", ""])) + str(int(rng.integers(0, 10**9))))
                    else: cols.append(str(int(rng.integers(10**9, 10**12))))
                elif c == 5 and rng.random() < 0.8:
                    cols.append(str(rng.choice(["+", "-", ".", "1", "-1", "+1", "--", ""])))
                else:
                    cols.append("".join(rng.choice(alphabet, size=int(rng.integers(0, 4)))))
            lines.append("\t".join(cols) + ("\t" if rng.random() < 0.1 else ""))
        f = tmp_path / ("t%d.bed" % trial)
        f.write_text("\n".join(lines) + "\n")
        args = [mode] + extra + ["-t", "3", "-c", "chr1,chr2,chrX", str(f)]
        fast = subprocess.run([TOOL] + args, capture_output=True)
        slow = subprocess.run([TOOL] + args, capture_output=True, env=dict(os.environ, GTX_NO_FAST_PARSE="1"))
        assert (fast.returncode, fast.stdout, fast.stderr) == (slow.returncode, slow.stdout, slow.stderr), lines


# ---- packed region files (.gtx): the text parse done once --------------------------------------------------
def _pack(src, dst):
    r = subprocess.run([TOOL, "pack", str(src), str(dst)], capture_output=True)
    return r.returncode, r.stderr.decode()


@pytest.mark.parametrize("mode,extra", [("ou", []), ("ou", ["-a"]), ("os", ["-z"]), ("os", ["-a", "-s", "-z"]), ("su", ["-a", "-l", "4"]),
                                        ("ss", []), ("ou", ["-l", "3"])])
def test_packed_file_gives_the_same_batches_as_its_text(tmp_path, mode, extra):
    rng = np.random.default_rng(17)
    names = ["chr1", "chr10", "chr2", "chrX", "chrUn_1"]
    n = 30000
    c = np.sort(rng.integers(0, len(names), size=n))
    s = np.concatenate([np.sort(rng.integers(0, 200000, size=int((c == k).sum()))) for k in range(len(names))])
    ln = rng.choice([0, 1, 36, 500], size=n)
    strand = rng.choice(["+", "-", "."], size=n)
    label = rng.choice(["5", "2", "abc", "-3", "0", "17"], size=n)
    order = np.lexsort((s, [names[k] for k in c]))                      # strcmp order of the names, then start
    f = tmp_path / "reads.bed"
    with open(f, "w") as h:
        h.write("track name=x\n")
        for i in order:
            h.write("%s\t%d\t%d\t%s\t0\t%s\n" % (names[c[i]], s[i], s[i] + ln[i], label[i], strand[i]))
    g = tmp_path / "reads.gtx"
    assert _pack(f, g) == (0, "")
    # the text path sees the header line only through GenomicRegionSet (which skips it): give the tool the body
    body = tmp_path / "body.bed"
    body.write_text("".join(open(f).readlines()[1:]))
    args = [mode] + extra + ["-t", "3", "-b", "7000", "-c", "chr1,chr10,chr2,chrX"]
    a = subprocess.run([TOOL] + args + [str(body)], capture_output=True)
    b = subprocess.run([TOOL] + args + [str(g)], capture_output=True)
    assert (a.returncode, a.stderr) == (b.returncode, b.stderr)
    # batches end at different places (text: blocks of lines, packed: record counts), so the per-batch "# zero" notes
    # interleave differently with the triples; the triples in order and the notes as a set must agree
    rows = lambda r: [l for l in r.stdout.decode().splitlines() if not l.startswith("#")]
    notes = lambda r: sorted(l for l in r.stdout.decode().splitlines() if l.startswith("#"))
    if "-a" in extra:                                    # strand-aware batches put their '-' reads behind their '+' reads
        assert sorted(rows(a)) == sorted(rows(b)) and notes(a) == notes(b)
    else:
        assert rows(a) == rows(b) and notes(a) == notes(b)
    assert len(a.stdout) > 1000 or a.returncode != 0


def test_packed_file_errors_and_order_checks(tmp_path):
    f = tmp_path / "u.bed"
    f.write_text("chr1\t10\t20\tx\t0\t+\nchr1\t5\t8\ty\t0\t-\nchr2\t1\t0\tz\t0\t+\n")
    g = tmp_path / "u.gtx"
    assert _pack(f, g) == (0, "")
    for mode, extra in (("os", []), ("ss", ["-a", "-s"]), ("ou", []), ("os", ["-s"])):
        a = subprocess.run([TOOL, mode] + extra + ["-c", "chr1,chr2", str(f)], capture_output=True)
        b = subprocess.run([TOOL, mode] + extra + ["-c", "chr1,chr2", str(g)], capture_output=True)
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr), (mode, extra)
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t10\t20\tx\t0\t+\nchr1\t5\n")
    rc, err = _pack(bad, tmp_path / "bad.gtx")
    assert rc == 1 and err == "\nError: Line 2: number of tokens should be at least 3 for BED format!\n"
    bad.write_text("chr1\t10\t20\tx\t0\t?\n")
    rc, err = _pack(bad, tmp_path / "bad.gtx")
    assert rc == 1 and err == "Error: invalid strand '?'!\n"
    (tmp_path / "junk.gtx").write_bytes(b"GTXP\x01\x00\x00\x00" + b"\xff" * 40)
    r = subprocess.run([TOOL, "ou", "-c", "chr1", str(tmp_path / "junk.gtx")], capture_output=True)
    assert r.returncode == 1 and b"not a valid packed region file" in r.stderr
